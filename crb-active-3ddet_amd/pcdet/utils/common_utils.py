"""Pure-torch geometry / bookkeeping helpers with the reference's names (pcdet/utils/common_utils.py)."""
import logging
import random

import numpy as np
import torch
import torch.distributed as dist


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def limit_period(val, offset=0.5, period=np.pi):
    """val - floor(val/period + offset)*period   (common_utils.py:24-27)"""
    val, is_numpy = check_numpy_to_torch(val)
    ans = val - torch.floor(val / period + offset) * period
    return ans.numpy() if is_numpy else ans


def rotate_points_along_z(points, angle):
    """points (B,N,3+C), angle (B) -> rotated about +z (common_utils.py:37-60)"""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    c, s = torch.cos(angle), torch.sin(angle)
    zeros, ones = torch.zeros_like(angle), torch.ones_like(angle)
    rot = torch.stack((c, s, zeros, -s, c, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    xyz = torch.matmul(points[:, :, 0:3], rot)
    out = torch.cat((xyz, points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def effective_cpu_count():
    """CPUs this process may actually use: the smallest of os.cpu_count(), the scheduler affinity mask and the cgroup CPU
    quota (a container on a 256-thread host with cpu.max = 16 cores gets 16: sizing worker pools by os.cpu_count() there
    makes 48 workers fight over 16 cores and starves the process that feeds the GPU)"""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(math.floor(int(q) / int(p)))))
    except (OSError, ValueError):
        try:                                                            # cgroup v1
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:
                q = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def mask_points_by_range(points, limit_range):
    return (points[:, 0] >= limit_range[0]) & (points[:, 0] <= limit_range[3]) & \
           (points[:, 1] >= limit_range[1]) & (points[:, 1] <= limit_range[4])


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N,3) [z,y,x] -> centers (N,3) xyz (common_utils.py:63-80)"""
    assert voxel_coords.shape[1] == 3
    if voxel_coords.is_cuda and voxel_coords.dtype == torch.int32 and voxel_coords.stride(1) == 1:
        # one launch (crb_voxel_centers) on the column slice itself; the scaled voxel size is the same f32 product as below
        import ctypes
        from crbhip import lib, check, ptr, cur_stream
        n = voxel_coords.shape[0]
        vs = (np.asarray(voxel_size, dtype=np.float32) * np.float32(downsample_times)).astype(np.float32)
        lo = np.asarray(point_cloud_range[0:3], dtype=np.float32)
        out = torch.empty((n, 3), dtype=torch.float32, device=voxel_coords.device)
        check(lib.crb_voxel_centers(ctypes.c_void_p(voxel_coords.data_ptr()) if n else None, max(3, voxel_coords.stride(0)), n,
                                    vs.ctypes.data_as(ctypes.c_void_p), lo.ctypes.data_as(ctypes.c_void_p), ptr(out),
                                    cur_stream(voxel_coords.device)), 'crb_voxel_centers')
        return out
    centers = voxel_coords.flip(1).float()          # (a python index list would be uploaded per call: a host sync)
    vs = device_constant(voxel_size, centers.device) * downsample_times
    pc_min = device_constant(point_cloud_range[0:3], centers.device)
    return (centers + 0.5) * vs + pc_min


_CONSTS = {}


def device_constant(values, device):
    """small f32 constant vector resident on `device` (cached: a torch.tensor(list, device=cuda) per call is a blocking
    pageable H2D copy, i.e. a host sync in the middle of the forward)"""
    key = (str(device), tuple(float(v) for v in values))
    t = _CONSTS.get(key)
    if t is None:
        if len(_CONSTS) > 256:
            _CONSTS.clear()
        t = _CONSTS[key] = torch.tensor(key[1], dtype=torch.float32, device=device)
    return t


_CHECK_SORTED = __import__('os').environ.get('CRB_CHECK_SORTED', '0') == '1'      # debug: one read-back per call


def batch_counts(bs_idx, batch_size):
    """rows per frame of a stacked tensor whose frame-index column is bs_idx -> (B) int32 (the reference's per-frame
    `(bs_idxs == k).sum()` loops: voxel_set_abstraction.py:321-323, pvrcnn_head.py:96-98). On the device: crb_sorted_key_counts, one
    launch of B waves that search the NON-DECREASING column (rows of a frame together, frames in order - what the counts mean to
    every stacked op). Host tensors: a scatter_add (torch.bincount would read back the maximum to size its output)."""
    if bs_idx.is_cuda:
        from crbhip import lib, check, ptr, cur_stream
        if _CHECK_SORTED and bs_idx.shape[0] > 1:          # CRB_CHECK_SORTED=1: an unsorted column would count wrong on the device only
            assert bool((bs_idx[1:] >= bs_idx[:-1]).all()), 'batch_counts: the frame-index column must be non-decreasing'
        if bs_idx.dtype not in (torch.float32, torch.int32):
            bs_idx = bs_idx.to(torch.int32)
        out = torch.empty((batch_size,), dtype=torch.int32, device=bs_idx.device)
        n = bs_idx.shape[0]
        import ctypes
        key = ctypes.c_void_p(bs_idx.data_ptr()) if n else None          # (a strided column: no copy)
        check(lib.crb_sorted_key_counts(key, int(bs_idx.dtype == torch.float32), max(1, bs_idx.stride(0)), n,
                                        int(batch_size), ptr(out), cur_stream(bs_idx.device)), 'crb_sorted_key_counts')
        return out
    if bs_idx.shape[0] > 1:      # host: the same contract as the device path, checked (free here), so the two cannot diverge silently
        assert bool((bs_idx[1:] >= bs_idx[:-1]).all()), 'batch_counts: the frame-index column must be non-decreasing'
    out = torch.zeros((batch_size,), dtype=torch.int32, device=bs_idx.device)
    return out.scatter_add_(0, bs_idx.long(), torch.ones_like(bs_idx, dtype=torch.int32))


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    logger = logging.getLogger(__name__)
    logger.setLevel(log_level if rank == 0 else 'ERROR')
    fmt = logging.Formatter('%(asctime)s  %(levelname)5s  %(message)s')
    console = logging.StreamHandler()
    console.setLevel(log_level if rank == 0 else 'ERROR')
    console.setFormatter(fmt)
    logger.addHandler(console)
    if log_file is not None:
        fh = logging.FileHandler(filename=log_file)
        fh.setLevel(log_level if rank == 0 else 'ERROR')
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    logger.propagate = False
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def keep_arrays_by_name(gt_names, used_classes):
    inds = [i for i, x in enumerate(gt_names) if x in used_classes]
    return np.array(inds, dtype=np.int64)


def init_dist_pytorch(tcp_port, local_rank, backend='nccl'):
    """one process per GPU; backend 'nccl' is RCCL on ROCm (common_utils.py:159-175)"""
    num_gpus = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(num_gpus, 1))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method='tcp://127.0.0.1:%d' % tcp_port,
                                rank=local_rank, world_size=num_gpus)
    return num_gpus, dist.get_rank()


def get_dist_info(return_gpu_per_machine=False):
    if dist.is_available() and dist.is_initialized():
        rank, world_size = dist.get_rank(), dist.get_world_size()
    else:
        rank, world_size = 0, 1
    if return_gpu_per_machine:
        return rank, world_size, max(torch.cuda.device_count(), 1)
    return rank, world_size


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
