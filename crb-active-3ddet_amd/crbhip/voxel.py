"""Host side of the HIP voxel generator (C-ABI: crb_voxelize)."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, host_i32x3, host_f32x3


def grid_size_xyz(point_cloud_range, voxel_size):
    """same rounding as pcdet/datasets/processor/data_processor.py:116-118"""
    import numpy as np
    r = np.asarray(point_cloud_range, dtype=np.float64)
    g = (r[3:6] - r[0:3]) / np.asarray(voxel_size, dtype=np.float64)
    return [int(v) for v in np.round(g).astype(np.int64)]


def voxelize(points, frame_offsets, point_cloud_range, voxel_size, max_voxels, max_points,
             want_voxels=True, want_mean=False, grid_xyz=None, lazy=False):
    """Batched point->voxel grouping on the GPU.

    points (n, C) f32 cuda, frames concatenated; frame_offsets (B+1) int32 cuda.
    Returns dict(voxels (M,max_points,C) | None, coords (M,4) i32 [b,z,y,x], num_points (M) i32,
                 mean (M,C) | None, counts (B) python list)  with M = sum of per-frame kept voxels.
    One host read-back of B+1 ints (the row count is data dependent).
    lazy=True: NO read-back - the tensors come back at their capacity together with `counts_dev` (B+1 i32 on the device, the last
    entry = M) and `pending`=True; the consumer (crbhip.sparse.build_rulebooks(..., n_dev=...)) reads M back together with its own
    counts and slices (one synchronisation per batch instead of two).
    """
    require_cuda(points, frame_offsets)
    assert points.dtype == torch.float32 and points.dim() == 2
    points = points.contiguous()
    frame_offsets = frame_offsets.to(torch.int32).contiguous()
    n, C = points.shape
    B = frame_offsets.numel() - 1
    if grid_xyz is None:
        grid_xyz = grid_size_xyz(point_cloud_range, voxel_size)
    cap = min(B * max_voxels, max(n, 1))
    dev = points.device
    voxels = torch.empty((cap, max_points, C), dtype=torch.float32, device=dev) if want_voxels else None
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num_points = torch.empty((cap,), dtype=torch.int32, device=dev)
    mean = torch.empty((cap, C), dtype=torch.float32, device=dev) if want_mean else None
    counts = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    ws_bytes = lib.crb_voxelize_workspace_bytes(n, B, max_voxels, max_points)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = lib.crb_voxelize(ptr(points), n, C, ptr(frame_offsets), B,
                          host_f32x3(point_cloud_range[0:3]), host_f32x3(voxel_size), host_i32x3(grid_xyz),
                          max_voxels, max_points, ptr(voxels), ptr(coords), ptr(num_points), ptr(mean), ptr(counts),
                          ptr(ws), ws_bytes, cur_stream(dev))
    check(rc, 'crb_voxelize')
    if lazy:
        return dict(voxels=voxels, coords=coords, num_points=num_points, mean=mean, counts=None, counts_dev=counts, pending=True)
    counts_h = counts.cpu().tolist()          # the one sync
    M = counts_h[-1]
    return dict(voxels=voxels[:M] if want_voxels else None, coords=coords[:M], num_points=num_points[:M],
                mean=mean[:M] if want_mean else None, counts=counts_h[:-1])
