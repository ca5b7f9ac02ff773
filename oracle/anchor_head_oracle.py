"""ORACLE — test infrastructure only. CPU restatement of the dense half of the SECOND step, independent of the product's
modules (only their PARAMETER TENSORS are read): anchors, per-frame / per-class target assignment, RPN losses and a
functional walk over the BEV backbone's layer list.

  anchors                pcdet/models/dense_heads/target_assigner/anchor_generator.py:17-60
  nearest-BEV IoU        pcdet/utils/box_utils.py:272-298 (boxes3d_nearest_bev_iou) / :236-269 (boxes_iou_normal)
  residual box encoding  pcdet/utils/box_coder_utils.py:13-43
  target assignment      pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-210
  RPN losses             pcdet/models/dense_heads/anchor_head_template.py:101-236, pcdet/utils/loss_utils.py:9-136
  BEV backbone           pcdet/models/backbones_2d/base_bev_backbone.py:81-112 (module order = the layer list :31-78)
  head                   pcdet/models/dense_heads/anchor_head_single.py:41-76
Pinned by tests/golden/ref_anchor_head.npz / ref_bev_vfe.npz (outputs of the reference's own classes)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def generate_anchors(anchor_range, anchor_cfgs, grid_sizes):
    """-> list over classes of (z=1, y, x, num_size, num_rot, 7) float32 tensors"""
    out = []
    r = np.asarray(anchor_range, dtype=np.float32)      # the reference keeps the range as a float32 ndarray (scalar maths follows)
    for cfg, gs in zip(anchor_cfgs, grid_sizes):
        sizes, rots, heights = cfg['anchor_sizes'], cfg['anchor_rotations'], cfg['anchor_bottom_heights']
        if cfg.get('align_center', False):
            xs, ys = (r[3] - r[0]) / gs[0], (r[4] - r[1]) / gs[1]
            xo, yo = xs / 2, ys / 2
        else:
            xs, ys = (r[3] - r[0]) / (gs[0] - 1), (r[4] - r[1]) / (gs[1] - 1)
            xo, yo = 0, 0
        x = torch.arange(r[0] + xo, r[3] + 1e-5, step=xs, dtype=torch.float32)
        y = torch.arange(r[1] + yo, r[4] + 1e-5, step=ys, dtype=torch.float32)
        z = torch.tensor(heights, dtype=torch.float32)
        a = torch.zeros((len(z), len(y), len(x), len(sizes), len(rots), 7), dtype=torch.float32)
        for zi in range(len(z)):
            for si in range(len(sizes)):
                for ri in range(len(rots)):
                    a[zi, :, :, si, ri, 0] = x[None, :]
                    a[zi, :, :, si, ri, 1] = y[:, None]
                    a[zi, :, :, si, ri, 2] = z[zi]
                    a[zi, :, :, si, ri, 3:6] = torch.tensor(sizes[si], dtype=torch.float32)
                    a[zi, :, :, si, ri, 6] = torch.tensor(rots[ri], dtype=torch.float32)
        a[..., 2] += a[..., 5] / 2
        out.append(a)
    return out


def _limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


def _nearest_bev(boxes):
    rot = torch.abs(_limit_period(boxes[:, 6], 0.5, np.pi))
    dims = torch.where((rot < np.pi / 4)[:, None], boxes[:, [3, 4]], boxes[:, [4, 3]])
    return torch.cat([boxes[:, :2] - dims / 2, boxes[:, :2] + dims / 2], 1)


def nearest_bev_iou(a, b):
    A, Bb = _nearest_bev(a), _nearest_bev(b)
    x_min = torch.max(A[:, 0, None], Bb[None, :, 0])
    x_max = torch.min(A[:, 2, None], Bb[None, :, 2])
    y_min = torch.max(A[:, 1, None], Bb[None, :, 1])
    y_max = torch.min(A[:, 3, None], Bb[None, :, 3])
    inter = torch.clamp_min(x_max - x_min, 0) * torch.clamp_min(y_max - y_min, 0)
    area_a = (A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1])
    area_b = (Bb[:, 2] - Bb[:, 0]) * (Bb[:, 3] - Bb[:, 1])
    return inter / torch.clamp_min(area_a[:, None] + area_b[None, :] - inter, 1e-6)


def residual_encode(boxes, anchors):
    anchors = anchors.clone()
    boxes = boxes.clone()
    anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], 1e-5)
    boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], 1e-5)
    xa, ya, za, dxa, dya, dza, ra = [anchors[:, i] for i in range(7)]
    xg, yg, zg, dxg, dyg, dzg, rg = [boxes[:, i] for i in range(7)]
    diag = torch.sqrt(dxa ** 2 + dya ** 2)
    return torch.stack([(xg - xa) / diag, (yg - ya) / diag, (zg - za) / dza, torch.log(dxg / dxa), torch.log(dyg / dya),
                        torch.log(dzg / dza), rg - ra], 1)


def assign_single(anchors, gt, gt_classes, matched, unmatched):
    """one frame, one anchor class (axis_aligned_target_assigner.py:132-210 with pos_fraction None)"""
    n = anchors.shape[0]
    labels = -torch.ones((n,), dtype=torch.int32)
    targets = torch.zeros((n, 7))
    if len(gt) == 0 or n == 0:
        labels[:] = 0
        return labels, targets, torch.zeros((n,))
    iou = nearest_bev_iou(anchors, gt)
    a2g = iou.argmax(1)
    a2g_max = iou[torch.arange(n), a2g]
    g2a = iou.argmax(0)
    g2a_max = iou[g2a, torch.arange(len(gt))]
    g2a_max[g2a_max == 0] = -1
    forced = (iou == g2a_max).nonzero()[:, 0]
    labels[forced] = gt_classes[a2g[forced]]
    pos = a2g_max >= matched
    labels[pos] = gt_classes[a2g[pos]]
    bg = (a2g_max < unmatched).nonzero()[:, 0]
    fg = (labels > 0).nonzero()[:, 0]
    labels[bg] = 0
    labels[forced] = gt_classes[a2g[forced]]
    targets[fg] = residual_encode(gt[a2g[fg]], anchors[fg])
    w = torch.zeros((n,))
    w[labels > 0] = 1.0
    return labels, targets, w


def assign_targets(anchors_list, gt_boxes_with_classes, class_names, anchor_cfgs):
    """-> box_cls_labels (B,A) int32, box_reg_targets (B,A,7), reg_weights (B,A); A ordered (y, x, class, rot) like the head"""
    class_names = np.array(class_names)
    B = gt_boxes_with_classes.shape[0]
    L, T, W = [], [], []
    for k in range(B):
        cur = gt_boxes_with_classes[k]
        cnt = len(cur) - 1
        while cnt > 0 and cur[cnt].sum() == 0:
            cnt -= 1
        cur = cur[:cnt + 1]
        cls = cur[:, -1].int()
        per_l, per_t, per_w = [], [], []
        for cfg, anchors in zip(anchor_cfgs, anchors_list):
            fm = anchors.shape[:3]
            mask = torch.tensor([class_names[int(c) - 1] == cfg['class_name'] for c in cls], dtype=torch.bool)
            l, t, w = assign_single(anchors.reshape(-1, 7), cur[mask][:, :7], cls[mask], cfg['matched_threshold'],
                                    cfg['unmatched_threshold'])
            per_l.append(l.view(*fm, -1))
            per_t.append(t.view(*fm, -1, 7))
            per_w.append(w.view(*fm, -1))
        L.append(torch.cat(per_l, -1).view(-1))
        T.append(torch.cat(per_t, -2).view(-1, 7))
        W.append(torch.cat(per_w, -1).view(-1))
    return torch.stack(L), torch.stack(T), torch.stack(W)


def focal_loss(logits, targets, weights, alpha=0.25, gamma=2.0):
    p = torch.sigmoid(logits)
    alpha_w = targets * alpha + (1 - targets) * (1 - alpha)
    pt = targets * (1.0 - p) + (1.0 - targets) * p
    bce = torch.clamp(logits, min=0) - logits * targets + torch.log1p(torch.exp(-torch.abs(logits)))
    return alpha_w * torch.pow(pt, gamma) * bce * weights.unsqueeze(-1)


def smooth_l1(pred, target, weights, beta=1.0 / 9.0):
    target = torch.where(torch.isnan(target), pred, target)
    n = torch.abs(pred - target)
    loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss * weights.unsqueeze(-1)


def rpn_loss(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors_list, num_class=3, dir_offset=0.78539,
             num_bins=2, w_cls=1.0, w_loc=2.0, w_dir=0.2):
    """cls_preds (B,H,W,A*C), box_preds (B,H,W,A*7), dir_preds (B,H,W,A*bins) -> total, cls, loc, dir"""
    B = cls_preds.shape[0]
    cared = labels >= 0
    positives = labels > 0
    negatives = labels == 0
    cls_w = (negatives * 1.0 + 1.0 * positives).float()
    norm = positives.sum(1, keepdim=True).float()
    cls_w = cls_w / torch.clamp(norm, min=1.0)
    reg_w = positives.float() / torch.clamp(norm, min=1.0)
    tgt = (labels * cared.type_as(labels)).long()
    one_hot = torch.zeros(*tgt.shape, num_class + 1)
    one_hot.scatter_(-1, tgt.unsqueeze(-1), 1.0)
    cls_loss = focal_loss(cls_preds.view(B, -1, num_class), one_hot[..., 1:], cls_w).sum() / B * w_cls
    bp = box_preds.view(B, -1, 7)
    sin_p = torch.sin(bp[..., 6:7]) * torch.cos(reg_targets[..., 6:7])
    sin_t = torch.cos(bp[..., 6:7]) * torch.sin(reg_targets[..., 6:7])
    loc = smooth_l1(torch.cat([bp[..., :6], sin_p], -1), torch.cat([reg_targets[..., :6], sin_t], -1), reg_w)
    loc_loss = loc.sum() / B * w_loc
    anchors = torch.cat(anchors_list, dim=-3).view(1, -1, 7).repeat(B, 1, 1)
    rot_gt = reg_targets[..., 6] + anchors[..., 6]
    off = rot_gt - dir_offset
    off = off - torch.floor(off / (2 * np.pi) + 0) * (2 * np.pi)
    dir_t = torch.clamp(torch.floor(off / (2 * np.pi / num_bins)).long(), 0, num_bins - 1)
    dw = positives.float()
    dw = dw / torch.clamp(dw.sum(-1, keepdim=True), min=1.0)
    logits = dir_preds.view(B, -1, num_bins)
    ce = F.cross_entropy(logits.permute(0, 2, 1), dir_t, reduction='none') * dw
    dir_loss = ce.sum() / B * w_dir
    return cls_loss + loc_loss + dir_loss, cls_loss, loc_loss, dir_loss


def run_layers(seq, x):
    """functional walk over an nn.Sequential of ZeroPad2d / Conv2d / ConvTranspose2d / BatchNorm2d / ReLU reading only
    the parameter tensors (train-mode batch statistics, no running-stat update)"""
    for m in seq:
        if isinstance(m, nn.ZeroPad2d):
            x = F.pad(x, m.padding)
        elif isinstance(m, nn.Conv2d):
            x = F.conv2d(x, m.weight, m.bias, m.stride, m.padding)
        elif isinstance(m, nn.ConvTranspose2d):
            x = F.conv_transpose2d(x, m.weight, m.bias, m.stride, m.padding)
        elif isinstance(m, nn.BatchNorm2d):
            x = F.batch_norm(x, None, None, m.weight, m.bias, True, 0.0, m.eps)
        elif isinstance(m, nn.ReLU):
            x = torch.relu(x)
        else:
            raise TypeError(type(m))
    return x


def bev_backbone(backbone, x):
    """base_bev_backbone.py:81-112: blocks in sequence, each followed by its up-sampling branch, concatenated"""
    ups = []
    for i, blk in enumerate(backbone.blocks):
        x = run_layers(blk, x)
        ups.append(run_layers(backbone.deblocks[i], x) if len(backbone.deblocks) > 0 else x)
    x = torch.cat(ups, 1) if len(ups) > 1 else ups[0]
    if len(backbone.deblocks) > len(backbone.blocks):
        x = run_layers(backbone.deblocks[-1], x)
    return x


def head_preds(head, feats):
    f = lambda conv: F.conv2d(feats, conv.weight, conv.bias).permute(0, 2, 3, 1).contiguous()
    return f(head.conv_cls), f(head.conv_box), f(head.conv_dir_cls)
