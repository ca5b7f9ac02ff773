"""AnchorHeadSingle (pcdet/models/dense_heads/anchor_head_single.py:7-76)."""
import numpy as np
import torch
import torch.nn as nn

from ...utils.linear_rows import LinearRows, rows_view
from .anchor_head_template import AnchorHeadTemplate


FUSED_HEAD_CONVS = True     # cls / box / dir 1x1 convs as one convolution over the concatenated filters
ROWS_GEMM = True            # ... run as a row GEMM on the channels_last map (utils/linear_rows.py)


class AnchorHeadSingle(AnchorHeadTemplate):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__(model_cfg=model_cfg, num_class=num_class, class_names=class_names, grid_size=grid_size,
                         point_cloud_range=point_cloud_range, predict_boxes_when_training=predict_boxes_when_training)
        self.num_anchors_per_location = sum(self.num_anchors_per_location)
        n = self.num_anchors_per_location
        self.conv_cls = nn.Conv2d(input_channels, n * self.num_class, kernel_size=1)
        self.conv_box = nn.Conv2d(input_channels, n * self.box_coder.code_size, kernel_size=1)
        if self.model_cfg.get('USE_DIRECTION_CLASSIFIER', None) is not None:
            self.conv_dir_cls = nn.Conv2d(input_channels, n * self.model_cfg.NUM_DIR_BINS, kernel_size=1)
        else:
            self.conv_dir_cls = None
        self.lazy_box_decode = False      # set by a detector whose RoI head reads the RPN boxes through proposal_layer only
        self.init_weights()

    def init_weights(self):
        pi = 0.01
        nn.init.constant_(self.conv_cls.bias, -np.log((1 - pi) / pi))
        nn.init.normal_(self.conv_box.weight, mean=0, std=0.001)

    def forward(self, data_dict):
        feats = data_dict['spatial_features_2d']
        if FUSED_HEAD_CONVS:
            # the three 1x1 convs (anchor_head_single.py:57-72) read the same (B,512,H,W) map — 1.15 GB at KITTI bs=16:
            # as ONE convolution over the concatenated filters the map is read once instead of three times, and its
            # gradient is written once instead of being summed from three grad_inputs (two 3.4 GB add passes per step).
            # The parameters stay the three modules' (same state_dict); outputs are the same numbers per channel.
            heads = [self.conv_cls, self.conv_box] + ([self.conv_dir_cls] if self.conv_dir_cls is not None else [])
            w = torch.cat([h.weight for h in heads], 0)
            b = torch.cat([h.bias for h in heads], 0)
            rows = rows_view(feats) if (ROWS_GEMM and feats.is_cuda) else None
            if rows is not None:
                # channels_last: the 1x1 convolution is the GEMM (B*H*W, 512) x (512, sum C); its output rows already are the
                # (B,H,W,sum C) layout the reference permutes to
                n, _, h, w_ = feats.shape
                y = LinearRows.apply(rows, w.flatten(1), b).view(n, h, w_, w.shape[0])
            else:
                y = torch.nn.functional.conv2d(feats, w, b).permute(0, 2, 3, 1)            # (B,H,W,sum C)
            outs = torch.split(y, [h.out_channels for h in heads], dim=3)
            cls_preds, box_preds = outs[0].contiguous(), outs[1].contiguous()
            dir_cls_preds = outs[2].contiguous() if self.conv_dir_cls is not None else None
        else:
            cls_preds = self.conv_cls(feats).permute(0, 2, 3, 1).contiguous()      # (B,H,W,C)
            box_preds = self.conv_box(feats).permute(0, 2, 3, 1).contiguous()
            dir_cls_preds = None
            if self.conv_dir_cls is not None:
                dir_cls_preds = self.conv_dir_cls(feats).permute(0, 2, 3, 1).contiguous()
        self.forward_ret_dict['cls_preds'] = cls_preds
        self.forward_ret_dict['box_preds'] = box_preds
        self.forward_ret_dict['dir_cls_preds'] = dir_cls_preds
        if self.training:
            self.forward_ret_dict.update(self.assign_targets(gt_boxes=data_dict['gt_boxes']))
        if not self.training or self.predict_boxes_when_training:
            B = data_dict['batch_size']
            data_dict['rpn_preds'] = cls_preds
            data_dict['cls_preds_normalized'] = False
            if self.lazy_box_decode:
                # a RoI head follows: its proposal layer decodes the top-k anchors it keeps (batch_box_decoder), the other
                # ~200,000 boxes per frame are never formed
                data_dict['batch_cls_preds'] = cls_preds.view(B, -1, self.num_class).float()
                data_dict['batch_box_decoder'] = lambda idx: self.generate_predicted_boxes(
                    batch_size=B, cls_preds=cls_preds, box_preds=box_preds, dir_cls_preds=dir_cls_preds, anchor_idx=idx)[1]
                data_dict.pop('batch_box_preds', None)
            else:
                batch_cls_preds, batch_box_preds = self.generate_predicted_boxes(
                    batch_size=B, cls_preds=cls_preds, box_preds=box_preds, dir_cls_preds=dir_cls_preds)
                data_dict['batch_cls_preds'] = batch_cls_preds
                data_dict['batch_box_preds'] = batch_box_preds
        return data_dict
