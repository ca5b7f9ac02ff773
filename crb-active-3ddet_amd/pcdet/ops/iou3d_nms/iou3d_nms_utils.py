"""pcdet.ops.iou3d_nms.iou3d_nms_utils (reference: pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-116) over the gfx950
kernels in csrc/iou3d_nms.hip. Same function names / argument meaning / return values:
  nms_gpu(boxes, scores, thresh, pre_maxsize=None) -> (LongTensor keep indices into the INPUT order, None)
Extra (no host synchronisation, fixed-size output): nms_gpu_padded, nms_batched."""
import torch

from crbhip import lib, check, ptr, cur_stream, require_cuda
from ...utils import common_utils


def _pairwise(boxes_a, boxes_b, mode):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    require_cuda(boxes_a, boxes_b)
    a = boxes_a.float().contiguous()
    b = boxes_b.float().contiguous()
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.crb_boxes_pairwise(ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out), mode, cur_stream(a.device)),
          'crb_boxes_pairwise')
    return out


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """host arrays in / host array out, computed by the same HIP kernel (the reference's CPU twin, iou3d_cpu.cpp:232-252)"""
    boxes_a, is_numpy = common_utils.check_numpy_to_torch(boxes_a)
    boxes_b, _ = common_utils.check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    out = _pairwise(boxes_a.cuda(), boxes_b.cuda(), 1).cpu()
    return out.numpy() if is_numpy else out


def boxes_overlap_bev(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, 0)


def boxes_iou_bev(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, 1)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    return _pairwise(boxes_a, boxes_b, 2)


def nms_batched(boxes_sorted, counts, thresh, max_keep, rotated=True):
    """boxes_sorted (B,N,7) in descending score order, counts (B) int32 or None
    -> keep (B,max_keep) int32 (-1 padded, indices into the sorted order), num_keep (B) int32. No host sync."""
    require_cuda(boxes_sorted)
    assert boxes_sorted.dim() == 3 and boxes_sorted.shape[2] == 7
    bs = boxes_sorted.float().contiguous()
    B, N = bs.shape[0], bs.shape[1]
    dev = bs.device
    keep = torch.empty((B, max_keep), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    wsb = lib.crb_nms_workspace_bytes(B, N)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    if counts is not None:
        counts = counts.to(torch.int32).contiguous()
    check(lib.crb_nms_batched(ptr(bs), ptr(counts), B, N, float(thresh), 1 if rotated else 0, int(max_keep), ptr(keep),
                              ptr(num), ptr(ws), wsb, cur_stream(dev)), 'crb_nms_batched')
    return keep, num


def _nms(boxes, scores, thresh, pre_maxsize, rotated):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    n = order.shape[0]
    if n == 0:
        return order, None
    keep, num = nms_batched(boxes[order][None], None, thresh, n, rotated)
    k = int(num.item())                      # the reference contract returns a variable-length index tensor
    return order[keep[0, :k].long()].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    return _nms(boxes, scores, thresh, pre_maxsize, True)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    return _nms(boxes, scores, thresh, None, False)


def nms_gpu_padded(boxes, scores, thresh, pre_maxsize, post_maxsize, rotated=True):
    """sync-free variant: -> (idx (post_maxsize) long into the input order, valid (post_maxsize) bool)"""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    if order.shape[0] == 0:
        z = torch.zeros((post_maxsize,), dtype=torch.long, device=boxes.device)
        return z, torch.zeros_like(z, dtype=torch.bool)
    keep, _ = nms_batched(boxes[order][None], None, thresh, post_maxsize, rotated)
    valid = keep[0] >= 0
    return order[keep[0].clamp(min=0).long()], valid
