"""MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31).

Two entry layouts:
  * reference layout: batch_dict['voxels'] (M,T,C), ['voxel_num_points'] (M) produced by the data loader ->
    masked mean exactly like the reference;
  * device-resident layout (MI355X path): batch_dict has raw 'points' (N,1+C) [b,x,y,z,..] sorted by frame and no
    'voxels' -> the HIP voxel generator (crb_voxelize) groups the points AND emits the per-voxel mean in the same pass,
    so the padded (M,T,C) tensor never exists and nothing but raw points crosses PCIe."""
import torch

from crbhip import voxel as _vx
from .vfe_template import VFETemplate

# The voxel count stays on the device until the 3-D backbone's table plan reads it back together with the sizes of its strided
# levels (one synchronisation per batch instead of two: crbhip.voxel.voxelize(lazy=True), crb_spconv_chain_mark_lazy). The batch
# then carries the generator's CAPACITY buffers + 'voxel_count_dev'; finish_lazy_voxels() gives any earlier consumer the cut tensors.
# Only on request of the caller ('_lazy_voxel_count' in the batch: the detector's module loop sets it when its 3-D backbone plans its
# tables): a direct call of the module returns cut tensors as the reference's does.
LAZY_VOXEL_COUNT = __import__('os').environ.get('CRB_LAZY_VOXEL_COUNT', '1') != '0'


def finish_lazy_voxels(batch_dict, n=None):
    """cut the capacity buffers of a lazily voxelized batch to their rows (n from the caller's read-back, or read it back here)"""
    cnt = batch_dict.pop('voxel_count_dev', None)
    if cnt is None:
        return batch_dict
    if n is None:
        n = int(cnt.cpu()[0])
    for k in ('voxel_coords', 'voxel_num_points', 'voxel_features'):
        batch_dict[k] = batch_dict[k][:n]
    return batch_dict


class MeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size=None, point_cloud_range=None, grid_size=None,
                 max_num_voxels=None, max_points_per_voxel=None, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features
        self.voxel_size = None if voxel_size is None else [float(v) for v in voxel_size]
        self.point_cloud_range = None if point_cloud_range is None else [float(v) for v in point_cloud_range]
        self.grid_size = None if grid_size is None else [int(v) for v in grid_size]
        # DATA_PROCESSOR.transform_points_to_voxels values (kitti_dataset.yaml:64-70) handed over by the dataset
        self.max_points_per_voxel = int(max_points_per_voxel or 5)
        self.max_voxels = dict(max_num_voxels or dict(train=16000, test=40000))

    def get_output_feature_dim(self):
        return self.num_point_features

    def _voxelize_on_device(self, batch_dict):
        pts = batch_dict['points']
        B = int(batch_dict['batch_size'])
        if 'point_frame_offsets' in batch_dict:
            off = batch_dict['point_frame_offsets']
        else:
            counts = torch.bincount(pts[:, 0].long(), minlength=B)
            off = torch.zeros(B + 1, dtype=torch.int32, device=pts.device)
            off[1:] = torch.cumsum(counts, 0).int()
        xyzf = pts[:, 1:].contiguous()
        mv = self.max_voxels['train' if self.training else 'test']
        r = _vx.voxelize(xyzf, off, self.point_cloud_range, self.voxel_size, mv, self.max_points_per_voxel,
                         want_voxels=False, want_mean=True, grid_xyz=self.grid_size, lazy=LAZY_VOXEL_COUNT and bool(batch_dict.get('_lazy_voxel_count', False)))
        if r.get('pending'):
            batch_dict['voxel_count_dev'] = r['counts_dev'][B:B + 1]
        batch_dict['voxel_coords'] = r['coords']
        batch_dict['voxel_num_points'] = r['num_points']
        batch_dict['voxel_features'] = r['mean']
        return batch_dict

    def forward(self, batch_dict, **kwargs):
        if 'voxels' not in batch_dict:
            return self._voxelize_on_device(batch_dict)
        voxel_features, voxel_num_points = batch_dict['voxels'], batch_dict['voxel_num_points']
        points_sum = voxel_features.sum(dim=1, keepdim=False)
        normalizer = torch.clamp_min(voxel_num_points.view(-1, 1), min=1.0).type_as(voxel_features)
        batch_dict['voxel_features'] = (points_sum / normalizer).contiguous()
        return batch_dict
