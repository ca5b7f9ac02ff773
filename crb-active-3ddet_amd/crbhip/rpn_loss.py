"""RPN losses of the anchor head as one HIP forward (+ finalize) and one HIP backward launch (csrc/rpn_loss.hip).

Host mirror of AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss (pcdet/models/dense_heads/
anchor_head_template.py:101-214) with SigmoidFocalClassificationLoss, WeightedSmoothL1Loss (sine difference of the heading) and
WeightedCrossEntropyLoss (pcdet/utils/loss_utils.py:9-188): `rpn_loss(...)` returns the (B,3) per-frame {cls, loc, dir} losses
(already multiplied by LOSS_WEIGHTS, normalised by the frame's positives); `.sum(0) / B` are the reference's three scalars."""
import ctypes

import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, CrbHipError


class RpnLossCfg(ctypes.Structure):
    """CrbRpnLossCfg of include/crb_hip.h"""
    _fields_ = [('alpha', ctypes.c_float), ('gamma', ctypes.c_float), ('beta', ctypes.c_float),
                ('dir_offset', ctypes.c_float), ('code_weights', ctypes.c_float * 7), ('cls_weight', ctypes.c_float),
                ('loc_weight', ctypes.c_float), ('dir_weight', ctypes.c_float), ('num_class', ctypes.c_int32),
                ('num_dir_bins', ctypes.c_int32)]


def make_cfg(num_class, num_dir_bins, code_weights, cls_weight, loc_weight, dir_weight, dir_offset, alpha=0.25, gamma=2.0,
             beta=1.0 / 9.0):
    cw = [float(x) for x in code_weights]
    if len(cw) != 7:
        raise CrbHipError('crb_rpn_loss: 7 code weights (code size 7) expected, got %d' % len(cw))
    return RpnLossCfg(alpha, gamma, beta, dir_offset, (ctypes.c_float * 7)(*cw), cls_weight, loc_weight, dir_weight,
                      int(num_class), int(num_dir_bins))


class _RpnLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, cfg):
        B, A = int(labels.shape[0]), int(labels.shape[1])
        dev = cls_preds.device
        loss = torch.empty((B, 3), dtype=torch.float32, device=dev)
        npos = torch.empty((B,), dtype=torch.float32, device=dev)
        wsb = lib.crb_rpn_loss_workspace_bytes(B, A)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        check(lib.crb_rpn_loss_forward(ptr(cls_preds), ptr(box_preds), ptr(dir_preds), ptr(labels), ptr(reg_targets),
                                       ptr(anchors), B, A, ctypes.byref(cfg), ptr(loss), ptr(npos), ptr(ws), wsb,
                                       cur_stream(dev)), 'crb_rpn_loss_forward')
        ctx.save_for_backward(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, npos)
        ctx.cfg = cfg
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, npos = ctx.saved_tensors
        B, A = int(labels.shape[0]), int(labels.shape[1])
        dev = cls_preds.device
        g = grad_loss.contiguous().float()
        d_cls = torch.empty_like(cls_preds)
        d_box = torch.empty_like(box_preds)
        d_dir = torch.empty_like(dir_preds) if dir_preds is not None else None
        check(lib.crb_rpn_loss_backward(ptr(cls_preds), ptr(box_preds), ptr(dir_preds), ptr(labels), ptr(reg_targets),
                                        ptr(anchors), B, A, ctypes.byref(ctx.cfg), ptr(npos), ptr(g), ptr(d_cls), ptr(d_box),
                                        ptr(d_dir), cur_stream(dev)), 'crb_rpn_loss_backward')
        return d_cls, d_box, d_dir, None, None, None, None


def rpn_loss(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors, cfg):
    """cls_preds (B,...,A_loc*num_class), box_preds (B,...,A_loc*7), dir_preds (B,...,A_loc*bins) or None — the head's
    (B,H,W,C) outputs, whose memory already is (B,A,.) —, labels (B,A) i32, reg_targets (B,A,7), anchors (A,7)
    -> (B,3) per-frame {cls, loc, dir} losses"""
    require_cuda(cls_preds, box_preds, dir_preds, labels, reg_targets, anchors)
    B, A = int(labels.shape[0]), int(labels.shape[1])
    if cls_preds.numel() != B * A * cfg.num_class or box_preds.numel() != B * A * 7 or reg_targets.numel() != B * A * 7 \
            or anchors.numel() != A * 7 or (dir_preds is not None and dir_preds.numel() != B * A * cfg.num_dir_bins):
        raise CrbHipError('crb_rpn_loss: tensor sizes do not match (B, A) = (%d, %d)' % (B, A))
    if labels.dtype != torch.int32:
        labels = labels.to(torch.int32)
    f = lambda t: None if t is None else t.contiguous().float()
    return _RpnLoss.apply(f(cls_preds), f(box_preds), f(dir_preds), labels.contiguous(), f(reg_targets), f(anchors), cfg)
