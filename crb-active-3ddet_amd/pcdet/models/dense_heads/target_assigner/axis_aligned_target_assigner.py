"""AxisAlignedTargetAssigner (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:8-210).

Same assignment rule, evaluated for the whole batch at once with no host synchronisation: the reference loops
frame x class in Python with .cpu()/nonzero() round trips; here every (frame, class) pair is a
slice of one masked (B, A, G) IoU tensor.

Rule per (frame, anchor class c) with that class's ground truths:
  forced   = anchor attains some gt's best overlap (and that best overlap is > 0)
  positive = forced or max-overlap >= matched_threshold          -> label c, regression target = argmax gt
  negative = max-overlap < unmatched_threshold and not forced     -> label 0
  otherwise ignored (-1); a frame with no gt of class c labels every anchor 0.
Only the POS_FRACTION < 0 (no sampling) branch used by every shipped SECOND / PV-RCNN config is implemented."""
import numpy as np
import torch

from ....utils import box_utils


class AxisAlignedTargetAssigner(object):
    def __init__(self, model_cfg, class_names, box_coder, match_height=False):
        super().__init__()
        anchor_generator_cfg = model_cfg.ANCHOR_GENERATOR_CONFIG
        anchor_target_cfg = model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = box_coder
        self.match_height = match_height
        self.class_names = np.array(class_names)
        self.anchor_class_names = [c['class_name'] for c in anchor_generator_cfg]
        self.pos_fraction = anchor_target_cfg.POS_FRACTION if anchor_target_cfg.POS_FRACTION >= 0 else None
        self.sample_size = anchor_target_cfg.SAMPLE_SIZE
        self.norm_by_num_examples = anchor_target_cfg.NORM_BY_NUM_EXAMPLES
        self.matched_thresholds = {c['class_name']: c['matched_threshold'] for c in anchor_generator_cfg}
        self.unmatched_thresholds = {c['class_name']: c['unmatched_threshold'] for c in anchor_generator_cfg}
        self.use_multihead = model_cfg.get('USE_MULTIHEAD', False)
        if self.pos_fraction is not None or self.use_multihead or self.match_height:
            raise NotImplementedError('only POS_FRACTION<0, single head, nearest-BEV matching are on the hot path')

    def assign_targets(self, all_anchors, gt_boxes_with_classes):
        """all_anchors: [(nz,ny,nx,S,R,7) per class]; gt (B,G,8) zero padded, last column class id 1..C
        -> box_cls_labels (B,A) int32, box_reg_targets (B,A,code), reg_weights (B,A); A ordered (z,y,x,class,size,rot)"""
        gt = gt_boxes_with_classes
        B, G = gt.shape[0], gt.shape[1]
        gt_boxes, gt_cls = gt[..., :-1], gt[..., -1]
        # valid = everything up to the last non-zero row (axis_aligned_target_assigner.py:54-58; row 0 always kept)
        nonzero = gt_boxes.sum(-1) != 0
        idx = torch.arange(G, device=gt.device).view(1, G)
        last = torch.where(nonzero, idx, torch.zeros_like(idx)).max(dim=1, keepdim=True)[0]
        valid = idx <= last
        labels_l, targets_l, weights_l = [], [], []
        fm_shape = all_anchors[0].shape[:3]
        for c_idx, (cname, anchors) in enumerate(zip(self.anchor_class_names, all_anchors)):
            cid = int(np.nonzero(self.class_names == cname)[0][0]) + 1
            a = anchors.reshape(-1, anchors.shape[-1])
            m = valid & (gt_cls.int() == cid)
            lab, tgt, w = self.assign_targets_batched(a, gt_boxes, m, cid, self.matched_thresholds[cname],
                                                      self.unmatched_thresholds[cname])
            labels_l.append(lab.view(B, *fm_shape, -1))
            targets_l.append(tgt.view(B, *fm_shape, -1, self.box_coder.code_size))
            weights_l.append(w.view(B, *fm_shape, -1))
        return {
            'box_cls_labels': torch.cat(labels_l, dim=-1).view(B, -1),
            'box_reg_targets': torch.cat(targets_l, dim=-2).view(B, -1, self.box_coder.code_size),
            'reg_weights': torch.cat(weights_l, dim=-1).view(B, -1),
        }

    def assign_targets_batched(self, anchors, gt_boxes, gt_mask, class_id, matched_threshold, unmatched_threshold):
        """anchors (A,7); gt_boxes (B,G,7); gt_mask (B,G) -> labels (B,A) i32, targets (B,A,code), reg_weights (B,A)"""
        B, G = gt_mask.shape
        A = anchors.shape[0]
        iou = box_utils.boxes3d_nearest_bev_iou(anchors[None, :, 0:7], gt_boxes[..., 0:7])      # (B,A,G)
        iou = torch.where(gt_mask[:, None, :], iou, iou.new_full((), -1.0))
        a2g_max, a2g_arg = iou.max(dim=2)                                                       # (B,A)
        g2a_max = iou.max(dim=1)[0]                                                             # (B,G)
        g2a_max = torch.where(gt_mask & (g2a_max > 0), g2a_max, g2a_max.new_full((), -2.0))
        forced = (iou == g2a_max[:, None, :]).any(dim=2)
        has_gt = gt_mask.any(dim=1, keepdim=True)
        fg = forced | (a2g_max >= matched_threshold)
        bg = (a2g_max < unmatched_threshold) & ~forced
        labels = torch.where(fg, class_id, torch.where(bg, 0, -1))
        labels = torch.where(has_gt, labels, torch.zeros_like(labels)).int()
        fg = fg & has_gt
        sel = torch.gather(gt_boxes, 1, a2g_arg[..., None].expand(B, A, gt_boxes.shape[-1]))
        targets = self.box_coder.encode_torch(sel, anchors[None].expand(B, A, anchors.shape[-1]))
        targets = torch.where(fg[..., None], targets, torch.zeros_like(targets))
        if self.norm_by_num_examples:
            num_examples = (labels >= 0).sum(dim=1, keepdim=True).clamp(min=1).float()
            reg_weights = fg.float() / num_examples
        else:
            reg_weights = fg.float()
        return labels, targets, reg_weights

    def assign_targets_single(self, anchors, gt_boxes, gt_classes, matched_threshold=0.6, unmatched_threshold=0.45):
        """reference-shaped single (frame, class) entry point (axis_aligned_target_assigner.py:132-210)"""
        g = gt_boxes.shape[0]
        if g == 0:
            gt_b = anchors.new_zeros((1, 1, gt_boxes.shape[-1] if gt_boxes.dim() == 2 else 7))
            mask = torch.zeros((1, 1), dtype=torch.bool, device=anchors.device)
            cid = 1
        else:
            gt_b, mask = gt_boxes[None], torch.ones((1, g), dtype=torch.bool, device=anchors.device)
            cid = int(gt_classes[0])
        lab, tgt, w = self.assign_targets_batched(anchors, gt_b, mask, cid, matched_threshold, unmatched_threshold)
        return {'box_cls_labels': lab[0], 'box_reg_targets': tgt[0], 'reg_weights': w[0]}
