"""`spconv.utils.Point2VoxelCPU3d` surface used by pcdet/datasets/processor/data_processor.py:15-60, backed by the
gfx950 voxel generator (crb_voxelize). Host arrays in, host arrays out — exactly what VoxelGeneratorWrapper expects;
the batched device-resident path is crbhip.voxel.voxelize."""
import numpy as np
import torch

from crbhip import voxel as _vx


class _TV(object):
    """minimal cumm.tensorview.Tensor look-alike: .numpy() / .numpy_view()"""

    def __init__(self, arr):
        self._a = arr

    def numpy(self):
        return self._a.copy()

    def numpy_view(self):
        return self._a


class Point2VoxelCPU3d(object):
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_voxels, max_num_points_per_voxel):
        self.vsize = [float(v) for v in vsize_xyz]
        self.coors_range = [float(v) for v in coors_range_xyz]
        self.num_point_features = int(num_point_features)
        self.max_num_voxels = int(max_num_voxels)
        self.max_num_points_per_voxel = int(max_num_points_per_voxel)
        self.grid_size = _vx.grid_size_xyz(self.coors_range, self.vsize)

    def point_to_voxel(self, pc, clear_voxels=True):
        pts = pc.numpy_view() if hasattr(pc, 'numpy_view') else np.asarray(pc)
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        assert pts.shape[1] == self.num_point_features
        dev = torch.device('cuda', torch.cuda.current_device())
        p = torch.from_numpy(pts).to(dev)
        off = torch.tensor([0, pts.shape[0]], dtype=torch.int32, device=dev)
        r = _vx.voxelize(p, off, self.coors_range, self.vsize, self.max_num_voxels, self.max_num_points_per_voxel,
                         want_voxels=True, want_mean=False, grid_xyz=self.grid_size)
        voxels = r['voxels'].cpu().numpy()
        coords = r['coords'][:, 1:].contiguous().cpu().numpy()
        num = r['num_points'].cpu().numpy()
        return _TV(voxels), _TV(coords), _TV(num)


Point2VoxelCPU3d.__doc__ = 'see module docstring'
