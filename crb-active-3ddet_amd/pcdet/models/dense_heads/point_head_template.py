"""PointHeadTemplate (pcdet/models/dense_heads/point_head_template.py:8-215), the parts PointHeadSimple uses.
Target assignment runs ONE batched points-in-boxes launch per box set instead of a per-frame loop (:78-92)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...ops.roiaware_pool3d import roiaware_pool3d_utils
from ...utils import loss_utils


# point labels and the focal classification loss (+ gradient) as HIP launches (csrc/point_head.hip); CRB_POINT_HEAD_FUSED=0 = the torch
# expressions below (A/B, and what the kernels are tested against)
FUSED = __import__('os').environ.get('CRB_POINT_HEAD_FUSED', '1') == '1'


class _PointFocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, labels, alpha, gamma, weight):
        from crbhip import lib, check, ptr, cur_stream
        p = preds.contiguous().float()
        n, C = p.shape
        buf = torch.empty((3,), dtype=torch.float32, device=p.device)
        d = torch.empty_like(p)
        check(lib.crb_point_focal_loss(ptr(p), ptr(labels.contiguous()), n, C, alpha, gamma, weight, ptr(buf), ptr(d), cur_stream(p.device)),
              'crb_point_focal_loss')
        ctx.save_for_backward(d)
        ctx.pshape = preds.shape
        pos = buf[1]
        ctx.mark_non_differentiable(pos)
        ctx.set_materialize_grads(False)
        return buf[2], pos

    @staticmethod
    def backward(ctx, g, _gp):
        if g is None:
            return None, None, None, None, None
        (d,) = ctx.saved_tensors
        return (d * g).view(ctx.pshape), None, None, None, None


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
        if FUSED and pts.is_cuda and gt_boxes.shape[1] > 0:
            # the label arithmetic behind the two point-in-box queries as one launch (csrc/point_head.hip)
            from crbhip import lib, check, ptr, cur_stream
            inner = roiaware_pool3d_utils.points_in_boxes_gpu(pts, gt_boxes[:, :, 0:7].contiguous())
            outer = roiaware_pool3d_utils.points_in_boxes_gpu(pts, extend_gt_boxes[:, :, 0:7].contiguous())
            labels = torch.empty((B * M,), dtype=torch.int64, device=pts.device)
            gt = gt_boxes.contiguous().float()
            check(lib.crb_point_labels(ptr(inner), ptr(outer), ptr(gt), B, M, int(gt.shape[1]), int(gt.shape[2]), int(self.num_class),
                                       ptr(labels), cur_stream(pts.device)), 'crb_point_labels')
            return {'point_cls_labels': labels, 'point_box_labels': None, 'point_part_labels': None}
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
        if FUSED and reduce and preds.is_cuda and type(self.cls_loss_func).__name__ == 'SigmoidFocalClassificationLoss' and \
                labels.dtype == torch.int64:
            # focal loss + its gradient as one launch (csrc/point_head.hip)
            loss, pos = _PointFocalLoss.apply(preds, labels, float(self.cls_loss_func.alpha), float(self.cls_loss_func.gamma),
                                              float(self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['point_cls_weight']))
            tb_dict = {} if tb_dict is None else tb_dict
            tb_dict.update({'point_loss_cls': loss.detach(), 'point_pos_num': pos})
            return loss, tb_dict
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
