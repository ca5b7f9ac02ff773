"""SparseConvTensor: same constructor / attributes / methods the reference touches
(pcdet/models/backbones_3d/spconv_backbone.py:141-146, pcdet/utils/spconv_utils.py:28-34,
pcdet/models/backbones_2d/map_to_bev/height_compression.py:20-24,
pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:385-386)."""
import torch

from crbhip import sparse as _sp


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None, indice_dict=None,
                 benchmark=False):
        assert features.dim() == 2 and indices.dim() == 2 and indices.shape[1] == len(spatial_shape) + 1
        assert indices.dtype == torch.int32, 'indices must be int32 [b,z,y,x]'
        self._features = features
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = indice_dict if indice_dict is not None else {}
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.frame_offsets = None          # host row offsets per frame (B+1), set by plan_indices(with_frame_offsets=True)

    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, val):
        # spconv 2.x forbids in-place feature assignment (pcdet/utils/spconv_utils.py:28-34 works around it);
        # we simply allow it.
        self._features = val

    def replace_feature(self, feature):
        new = SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.grid, self.voxel_num,
                               self.indice_dict, self.benchmark)
        new.frame_offsets = self.frame_offsets
        return new

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        shape3 = self.spatial_shape
        idx = self.indices
        if len(shape3) == 2:     # 2-D tensors ride the 3-D kernel with D=1
            shape3 = [1] + shape3
            idx = torch.cat([idx[:, :1], torch.zeros_like(idx[:, :1]), idx[:, 1:]], dim=1).contiguous()
        out = _sp.to_dense(self.features, idx, self.batch_size, shape3)
        if len(self.spatial_shape) == 2:
            out = out[:, :, 0]
        if not channels_first:
            perm = [0] + list(range(2, out.dim())) + [1]
            out = out.permute(*perm).contiguous()
        return out

    @property
    def sparity(self):
        return self.indices.shape[0] / float(self.spatial_size * self.batch_size)
