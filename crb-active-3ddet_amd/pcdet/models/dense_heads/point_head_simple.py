"""PointHeadSimple (pcdet/models/dense_heads/point_head_simple.py:7-91): keypoint foreground segmentation for PV-RCNN."""
import torch

from ...utils import box_utils
from .point_head_template import PointHeadTemplate
from ...utils.fc_rows import fc_rows


class PointHeadSimple(PointHeadTemplate):
    def __init__(self, num_class, input_channels, model_cfg, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class)
        self.cls_layers = self.make_fc_layers(fc_cfg=self.model_cfg.CLS_FC, input_channels=input_channels,
                                              output_channels=num_class)

    def assign_targets(self, input_dict):
        gt_boxes = input_dict['gt_boxes']
        B = gt_boxes.shape[0]
        extend = box_utils.enlarge_box3d(gt_boxes.view(-1, gt_boxes.shape[-1]),
                                         extra_width=self.model_cfg.TARGET_CONFIG.GT_EXTRA_WIDTH).view(B, -1,
                                                                                                        gt_boxes.shape[-1])
        return self.assign_stack_targets(points=input_dict['point_coords'], gt_boxes=gt_boxes, extend_gt_boxes=extend,
                                         set_ignore_flag=True, use_ball_constraint=False, ret_part_labels=False)

    def get_loss(self, tb_dict=None, reduce=True):
        tb_dict = {} if tb_dict is None else tb_dict
        loss, tb1 = self.get_cls_layer_loss(reduce=reduce)
        tb_dict.update(tb1)
        return loss, tb_dict

    def forward(self, batch_dict):
        if self.model_cfg.get('USE_POINT_FEATURES_BEFORE_FUSION', False):
            feats = batch_dict['point_features_before_fusion']
        else:
            feats = batch_dict['point_features']
        preds = fc_rows(self.cls_layers, feats)
        ret = {'point_cls_preds': preds}
        batch_dict['point_cls_scores'], _ = torch.sigmoid(preds).max(dim=-1)
        if self.training:
            ret['point_cls_labels'] = self.assign_targets(batch_dict)['point_cls_labels']
        self.forward_ret_dict = ret
        return batch_dict
