"""pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils (reference: roiaware_pool3d_utils.py:9-107) over
csrc/roiaware_pool3d.hip."""
import torch
import torch.nn as nn
from torch.autograd import Function

from crbhip import lib, check, ptr, cur_stream, require_cuda
from ...utils import common_utils


def points_in_boxes_gpu(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32 index of the first containing box, -1 = background"""
    assert boxes.shape[0] == points.shape[0] and boxes.shape[2] == 7 and points.shape[2] == 3
    require_cuda(points, boxes)
    B, M, _ = points.shape
    out = torch.empty((B, M), dtype=torch.int32, device=points.device)
    check(lib.crb_points_in_boxes(B, boxes.shape[1], M, ptr(boxes.contiguous().float()),
                                  ptr(points.contiguous().float()), ptr(out), cur_stream(points.device)),
          'crb_points_in_boxes')
    return out


def points_in_boxes_cpu(points, boxes):
    """points (P,3), boxes (N,7) host arrays -> (N,P) int membership matrix (roiaware_pool3d_utils.py:9-25).
    NOTE: the reference's CPU twin uses a 1e-2 margin (roiaware_pool3d.cpp:131) while its GPU kernel uses 1e-5; this
    entry point is answered by the GPU kernel's rule, one box at a time semantics (every containing box is marked)."""
    points, is_numpy = common_utils.check_numpy_to_torch(points)
    boxes, _ = common_utils.check_numpy_to_torch(boxes)
    assert boxes.shape[1] == 7 and points.shape[1] == 3
    N, P = boxes.shape[0], points.shape[0]
    out = torch.zeros((N, P), dtype=torch.int32)
    if N and P:
        pts = points.float().cuda().view(1, P, 3).expand(N, P, 3).contiguous()
        idx = points_in_boxes_gpu(pts, boxes.float().cuda().view(N, 1, 7))
        out = (idx == 0).int().cpu()
    return out.numpy() if is_numpy else out


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size = out_size
        self.max_pts_each_voxel = max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        assert pool_method in ['max', 'avg']
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)


class RoIAwarePool3dFunction(Function):
    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        """rois (N,7), pts (P,3), pts_feature (P,C) -> (N,ox,oy,oz,C)"""
        require_cuda(rois, pts, pts_feature)
        assert rois.shape[1] == 7 and pts.shape[1] == 3
        if isinstance(out_size, int):
            ox = oy = oz = out_size
        else:
            ox, oy, oz = [int(v) for v in out_size]
        N, P, C = rois.shape[0], pts.shape[0], pts_feature.shape[1]
        dev = rois.device
        pooled = torch.zeros((N, ox, oy, oz, C), dtype=torch.float32, device=dev)
        argmax = torch.zeros((N, ox, oy, oz, C), dtype=torch.int32, device=dev)
        pts_idx = torch.zeros((N, ox, oy, oz, max_pts_each_voxel), dtype=torch.int32, device=dev)
        method = {'max': 0, 'avg': 1}[pool_method]
        check(lib.crb_roiaware_pool3d_forward(N, P, C, max_pts_each_voxel, ox, oy, oz, ptr(rois.contiguous().float()),
                                              ptr(pts.contiguous().float()), ptr(pts_feature.contiguous().float()),
                                              ptr(argmax), ptr(pts_idx), ptr(pooled), method, cur_stream(dev)),
              'crb_roiaware_pool3d_forward')
        ctx.roiaware_pool3d_for_backward = (pts_idx, argmax, method, P, C, (ox, oy, oz), max_pts_each_voxel)
        return pooled

    @staticmethod
    def backward(ctx, grad_out):
        pts_idx, argmax, method, P, C, (ox, oy, oz), mp = ctx.roiaware_pool3d_for_backward
        g = grad_out.contiguous().float()
        grad_in = torch.zeros((P, C), dtype=torch.float32, device=g.device)
        check(lib.crb_roiaware_pool3d_backward(pts_idx.shape[0], C, mp, ox, oy, oz, ptr(pts_idx), ptr(argmax), ptr(g),
                                               ptr(grad_in), method, cur_stream(g.device)),
              'crb_roiaware_pool3d_backward')
        return None, None, grad_in, None, None, None
