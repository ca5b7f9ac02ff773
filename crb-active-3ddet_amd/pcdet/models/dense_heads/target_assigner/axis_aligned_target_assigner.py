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

from crbhip import lib, check, ptr, cur_stream
from ....utils import box_utils

FUSED = True      # use the two-launch HIP assigner (crb_assign_targets) for device tensors; False = batched torch ops


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
        if FUSED and gt.is_cuda and self.box_coder.code_size == 7 and not self.norm_by_num_examples:
            return self.assign_targets_fused(all_anchors, gt)
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

    def _fused_constants(self, all_anchors):
        key = (all_anchors[0].device, all_anchors[0].data_ptr())
        if getattr(self, '_fused_key', None) != key:
            dev = all_anchors[0].device
            flat = torch.cat(all_anchors, dim=-3)                       # (nz,ny,nx, sum S, R, 7)
            per = [a.shape[3] * a.shape[4] for a in all_anchors]
            ids = [int(np.nonzero(self.class_names == n)[0][0]) + 1 for n in self.anchor_class_names]
            pattern = torch.tensor(sum([[i] * (a.shape[4]) * a.shape[3] for i, a in zip(ids, all_anchors)], []),
                                   dtype=torch.int32)
            # flattened order is (.., class*size, rot): class id of slot s = ids[s // R] for equal R per class
            R = all_anchors[0].shape[4]
            slots = torch.tensor(sum([[i] * a.shape[3] for i, a in zip(ids, all_anchors)], []), dtype=torch.int32)
            cls_of = slots.repeat_interleave(R)                          # (sum S * R)
            A = flat.numel() // 7
            anchor_cls = cls_of.repeat(A // cls_of.numel()).to(dev)
            nmax = max(ids) + 1
            m = torch.zeros(nmax, dtype=torch.float32)
            u = torch.zeros(nmax, dtype=torch.float32)
            for i, n in zip(ids, self.anchor_class_names):
                m[i] = self.matched_thresholds[n]
                u[i] = self.unmatched_thresholds[n]
            self._fused = (flat.reshape(-1, 7).contiguous(), anchor_cls.contiguous(), m.to(dev), u.to(dev))
            self._fused_key = key
        return self._fused

    def assign_targets_fused(self, all_anchors, gt):
        """same outputs as the batched torch path, two HIP launches (csrc/target_assign.hip)"""
        anchors, anchor_cls, m, u = self._fused_constants(all_anchors)
        B, G = gt.shape[0], gt.shape[1]
        A = anchors.shape[0]
        dev = gt.device
        gtc = gt.contiguous().float()
        nonzero = gtc[..., :-1].sum(-1) != 0
        idx = torch.arange(G, device=dev).view(1, G)
        last = torch.where(nonzero, idx, torch.zeros_like(idx)).max(dim=1, keepdim=True)[0]
        valid = (idx <= last).to(torch.uint8).contiguous()
        labels = torch.empty((B, A), dtype=torch.int32, device=dev)
        targets = torch.empty((B, A, 7), dtype=torch.float32, device=dev)
        weights = torch.empty((B, A), dtype=torch.float32, device=dev)
        wsb = lib.crb_assign_targets_workspace_bytes(B, A, G)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        check(lib.crb_assign_targets(ptr(anchors), ptr(anchor_cls), A, ptr(gtc), ptr(valid), B, G, ptr(m), ptr(u),
                                     ptr(labels), ptr(targets), ptr(weights), ptr(ws), wsb, cur_stream(dev)),
              'crb_assign_targets')
        return {'box_cls_labels': labels, 'box_reg_targets': targets, 'reg_weights': weights}

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
