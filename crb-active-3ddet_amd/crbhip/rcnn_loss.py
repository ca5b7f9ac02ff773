"""Second-stage losses of the RoI head as one HIP launch, gradients included, and the canonical transformation of the sampled
ground truths as one launch (csrc/rcnn_loss.hip).

Host mirror of RoIHeadTemplate.get_box_cls_layer_loss / get_box_reg_layer_loss (pcdet/models/roi_heads/roi_head_template.py:142-285,
BinaryCrossEntropy + smooth-l1 + corner regularisation, reduce=True, no `reg_sample_targets`) and of the tail of assign_targets
(:118-138)."""
import ctypes

import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, CrbHipError


class RcnnLossCfg(ctypes.Structure):
    """CrbRcnnLossCfg of include/crb_hip.h"""
    _fields_ = [('beta', ctypes.c_float), ('code_weights', ctypes.c_float * 7), ('cls_weight', ctypes.c_float),
                ('reg_weight', ctypes.c_float), ('corner_weight', ctypes.c_float), ('corner', ctypes.c_int32)]


def make_cfg(code_weights, cls_weight, reg_weight, corner_weight, corner, beta=1.0 / 9.0):
    cw = [float(x) for x in code_weights]
    if len(cw) != 7:
        raise CrbHipError('crb_rcnn_loss: 7 code weights (code size 7) expected, got %d' % len(cw))
    return RcnnLossCfg(float(beta), (ctypes.c_float * 7)(*cw), float(cls_weight), float(reg_weight), float(corner_weight),
                       1 if corner else 0)


class _RcnnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rcnn_cls, rcnn_reg, labels, reg_valid, rois, gt_local, gt_src, cfg):
        n = int(rcnn_reg.shape[0])
        dev = rcnn_reg.device
        buf = torch.empty((7,), dtype=torch.float32, device=dev)
        out, total = buf[:6], buf[6]                      # two disjoint views: the parts (no graph) and the scalar with the graph
        d_cls = torch.empty((n,), dtype=torch.float32, device=dev)
        d_reg = torch.empty((n, 7), dtype=torch.float32, device=dev)
        tgt = torch.empty((n, 7), dtype=torch.float32, device=dev)
        check(lib.crb_rcnn_loss(ptr(rcnn_cls), ptr(rcnn_reg), ptr(labels), int(labels.dtype == torch.int64), ptr(reg_valid), ptr(rois),
                                ptr(gt_local), ptr(gt_src), int(gt_local.shape[1]), n, ctypes.byref(cfg), ptr(buf), ptr(d_cls),
                                ptr(d_reg), ptr(tgt), cur_stream(dev)), 'crb_rcnn_loss')
        ctx.save_for_backward(d_cls, d_reg)
        ctx.shapes = (rcnn_cls.shape, rcnn_reg.shape)
        ctx.mark_non_differentiable(out, tgt)
        ctx.set_materialize_grads(False)
        return total, out, tgt

    @staticmethod
    def backward(ctx, g, _go, _gt):
        if g is None:
            return (None,) * 8
        d_cls, d_reg = ctx.saved_tensors
        return (d_cls * g).view(ctx.shapes[0]), (d_reg * g).view(ctx.shapes[1]), None, None, None, None, None, None


def rcnn_loss(rcnn_cls, rcnn_reg, cls_labels, reg_valid_mask, rois, gt_of_rois, gt_of_rois_src, cfg):
    """rcnn_cls (n,1) / (n), rcnn_reg (n,7), cls_labels (B,P) f32 or i64, reg_valid_mask (B,P) i64, rois (B,P,7),
    gt_of_rois / gt_of_rois_src (B,P,7+) -> (rcnn_loss scalar with the graph, parts (6) = {cls, reg, corner, total, fg, valid} detached,
    reg_targets (n,7))"""
    require_cuda(rcnn_cls, rcnn_reg, cls_labels, reg_valid_mask, rois, gt_of_rois, gt_of_rois_src)
    n = int(rcnn_reg.shape[0])
    if rcnn_reg.shape[-1] != 7 or rcnn_cls.numel() != n or cls_labels.numel() != n or reg_valid_mask.numel() != n or \
            rois.numel() != n * 7 or gt_of_rois.numel() != gt_of_rois_src.numel() or gt_of_rois.shape[-1] < 7:
        raise CrbHipError('crb_rcnn_loss: tensor sizes do not match n = %d RoIs of code size 7' % n)
    if cls_labels.dtype not in (torch.float32, torch.int64):
        cls_labels = cls_labels.float()
    if reg_valid_mask.dtype != torch.int64:
        reg_valid_mask = reg_valid_mask.long()
    c = gt_of_rois.shape[-1]
    f = lambda t: t.contiguous().float()
    return _RcnnLoss.apply(f(rcnn_cls).view(n), f(rcnn_reg), cls_labels.contiguous().view(n), reg_valid_mask.contiguous().view(n),
                           f(rois).view(n, 7), f(gt_of_rois).view(n, c), f(gt_of_rois_src).view(n, c), cfg)


@torch.no_grad()
def roi_canonical_targets(rois, gt_of_rois):
    """rois (B,P,7+), gt_of_rois (B,P,7+C) in LiDAR coordinates -> the same boxes in the RoI frame, heading folded into
    [-pi/2, pi/2] (roi_head_template.py:118-138)"""
    require_cuda(rois, gt_of_rois)
    rois, gt = rois.contiguous().float(), gt_of_rois.contiguous().float()
    n = gt.numel() // gt.shape[-1]
    if rois.numel() // rois.shape[-1] != n or rois.shape[-1] < 7 or gt.shape[-1] < 7:
        raise CrbHipError('crb_roi_canonical_targets: (.., 7+) boxes for the same RoIs expected')
    out = torch.empty_like(gt)
    check(lib.crb_roi_canonical_targets(ptr(rois), int(rois.shape[-1]), ptr(gt), int(gt.shape[-1]), n, ptr(out), cur_stream(gt.device)),
          'crb_roi_canonical_targets')
    return out


class RoiSamplerCfg(ctypes.Structure):
    """CrbRoiSamplerCfg of include/crb_hip.h"""
    _fields_ = [('roi_per_image', ctypes.c_int32), ('fg_quota', ctypes.c_int32), ('by_class', ctypes.c_int32),
                ('score_type', ctypes.c_int32), ('fg_thresh', ctypes.c_float), ('reg_fg_thresh', ctypes.c_float),
                ('cls_fg_thresh', ctypes.c_float), ('cls_bg_thresh', ctypes.c_float), ('cls_bg_thresh_lo', ctypes.c_float),
                ('hard_bg_ratio', ctypes.c_float), ('soft_den', ctypes.c_float)]


MAX_PROPOSALS = 1024            # one workgroup per frame, one thread per proposal


@torch.no_grad()
def roi_sample_targets(rois, roi_scores, roi_labels, gt_boxes, iou, u_perm, u_slot, cfg):
    """rois (B,R,7+), roi_scores (B,R), roi_labels (B,R) i64, gt_boxes (B,G,8+), iou (B*R, B*G), u_perm (B,R), u_slot (B,P)
    -> dict(sampled, rois, gt_of_rois, gt_iou_of_rois, roi_scores, roi_labels, reg_valid_mask, rcnn_cls_labels) of
    ProposalTargetLayer.forward (proposal_target_layer.py:15-61)"""
    require_cuda(rois, roi_scores, roi_labels, gt_boxes, iou, u_perm, u_slot)
    B, R, G, P = int(rois.shape[0]), int(rois.shape[1]), int(gt_boxes.shape[1]), int(cfg.roi_per_image)
    if roi_scores.shape != (B, R) or roi_labels.shape != (B, R) or iou.shape != (B * R, B * G) or u_perm.shape != (B, R) or \
            u_slot.shape != (B, P) or roi_labels.dtype != torch.int64:
        raise CrbHipError('crb_roi_sample_targets: shapes do not match (B, R, G, P) = (%d, %d, %d, %d)' % (B, R, G, P))
    dev = rois.device
    f = lambda t: t.contiguous().float()
    rois, gt = f(rois), f(gt_boxes)
    sampled = torch.empty((B, P), dtype=torch.int64, device=dev)
    o_rois = torch.empty((B, P, rois.shape[-1]), dtype=torch.float32, device=dev)
    o_gt = torch.empty((B, P, gt.shape[-1]), dtype=torch.float32, device=dev)
    o_iou = torch.empty((B, P), dtype=torch.float32, device=dev)
    o_scores = torch.empty((B, P), dtype=torch.float32, device=dev)
    o_labels = torch.empty((B, P), dtype=torch.int64, device=dev)
    valid = torch.empty((B, P), dtype=torch.int64, device=dev)
    cls = torch.empty((B, P), dtype=torch.float32 if cfg.score_type == 0 else torch.int64, device=dev)
    check(lib.crb_roi_sample_targets(ptr(rois), int(rois.shape[-1]), ptr(f(roi_scores)), ptr(roi_labels.contiguous()), ptr(gt),
                                     int(gt.shape[-1]), ptr(f(iou)), ptr(f(u_perm)), ptr(f(u_slot)), B, R, G, ctypes.byref(cfg),
                                     ptr(sampled), ptr(o_rois), ptr(o_gt), ptr(o_iou), ptr(o_scores), ptr(o_labels), ptr(valid),
                                     ptr(cls), cur_stream(dev)), 'crb_roi_sample_targets')
    return {'sampled': sampled, 'rois': o_rois, 'gt_of_rois': o_gt, 'gt_iou_of_rois': o_iou, 'roi_scores': o_scores,
            'roi_labels': o_labels, 'reg_valid_mask': valid, 'rcnn_cls_labels': cls}
