"""PointHeadTemplate (pcdet/models/dense_heads/point_head_template.py:8-215), the parts PointHeadSimple uses.
Target assignment runs ONE batched points-in-boxes launch per box set instead of a per-frame loop (:78-92)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...ops.roiaware_pool3d import roiaware_pool3d_utils
from ...utils import loss_utils


class PointHeadTemplate(nn.Module):
    def __init__(self, model_cfg, num_class):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.build_losses(self.model_cfg.LOSS_CONFIG)
        self.forward_ret_dict = None

    def build_losses(self, losses_cfg):
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        reg = losses_cfg.get('LOSS_REG', None)
        if reg == 'l1':
            self.reg_loss_func = F.l1_loss
        elif reg == 'WeightedSmoothL1Loss':
            self.reg_loss_func = loss_utils.WeightedSmoothL1Loss(
                code_weights=losses_cfg.LOSS_WEIGHTS.get('code_weights', None))
        else:
            self.reg_loss_func = F.smooth_l1_loss

    @staticmethod
    def make_fc_layers(fc_cfg, input_channels, output_channels):
        layers, c_in = [], input_channels
        for c in fc_cfg:
            layers += [nn.Linear(c_in, c, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            c_in = c
        layers.append(nn.Linear(c_in, output_channels, bias=True))
        return nn.Sequential(*layers)

    def assign_stack_targets(self, points, gt_boxes, extend_gt_boxes=None, ret_box_labels=False, ret_part_labels=False,
                             set_ignore_flag=True, use_ball_constraint=False, central_radius=2.0):
        """points (N,4) [b,x,y,z], frame-sorted, the same count per frame; gt_boxes (B,M,8)
        -> point_cls_labels (N) long: class (or 1) inside a gt box, -1 in the enlarged shell only, 0 elsewhere"""
        assert points.dim() == 2 and points.shape[1] == 4 and gt_boxes.dim() == 3 and gt_boxes.shape[2] == 8
        assert set_ignore_flag and not use_ball_constraint and not ret_box_labels and not ret_part_labels, \
            'only the PointHeadSimple target mode is on the hot path'
        B = gt_boxes.shape[0]
        bs = points[:, 0].long()
        M = int(points.shape[0] // B)
        if points.shape[0] != M * B:
            raise NotImplementedError('ragged keypoint counts: pad to a dense (B,M,3) tensor first')
        pts = points[:, 1:4].reshape(B, M, 3).contiguous()
        inner = roiaware_pool3d_utils.points_in_boxes_gpu(pts, gt_boxes[:, :, 0:7].contiguous()).long().view(-1)
        outer = roiaware_pool3d_utils.points_in_boxes_gpu(pts, extend_gt_boxes[:, :, 0:7].contiguous()).view(-1)
        fg = inner >= 0
        ignore = fg ^ (outer >= 0)
        labels = torch.zeros_like(inner)
        labels = torch.where(ignore, torch.full_like(labels, -1), labels)
        if self.num_class == 1:
            fg_val = torch.ones_like(labels)
        else:
            fg_val = gt_boxes[bs, inner.clamp(min=0), -1].long()
        labels = torch.where(fg, fg_val, labels)
        return {'point_cls_labels': labels, 'point_box_labels': None, 'point_part_labels': None}

    def get_cls_layer_loss(self, tb_dict=None, reduce=True):
        labels = self.forward_ret_dict['point_cls_labels'].view(-1)
        preds = self.forward_ret_dict['point_cls_preds'].view(-1, self.num_class)
        positives = labels > 0
        cls_weights = ((labels == 0) * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(dim=0).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        one_hot = preds.new_zeros(*labels.shape, self.num_class + 1)
        one_hot.scatter_(-1, (labels * (labels >= 0).long()).unsqueeze(-1).long(), 1.0)
        src = self.cls_loss_func(preds, one_hot[..., 1:], weights=cls_weights)
        loss = src.sum() if reduce else src.view(-1, self.model_cfg.NUM_KEYPOINTS).sum(-1)
        loss = loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['point_cls_weight']
        tb_dict = {} if tb_dict is None else tb_dict
        tb_dict.update({'point_loss_cls': (loss if reduce else loss[0]).detach(), 'point_pos_num': pos_normalizer.detach()})
        return loss, tb_dict

    def forward(self, **kwargs):
        raise NotImplementedError
