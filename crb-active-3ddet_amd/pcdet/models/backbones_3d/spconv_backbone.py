"""VoxelBackBone8x / VoxelResBackBone8x (pcdet/models/backbones_3d/spconv_backbone.py:69-293) on the gfx950 sparse-conv
kernels. Module / parameter names match the reference so its checkpoints load unchanged."""
from functools import partial

import torch
import torch.nn as nn

from ...utils.spconv_utils import replace_feature, spconv
from spconv.pytorch.conv import plan_indices

PLAN_INDICES = True      # A/B: False builds every rulebook where the forward pass first needs it (a host sync per strided layer)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None):
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=True,
                                       indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True,
                                       indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = replace_feature(out, self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out = replace_feature(out, self.bn2(out.features))
        if self.downsample is not None:
            identity = self.downsample(x)
        return replace_feature(out, self.relu(out.features + identity.features))


class _Backbone8xBase(nn.Module):
    ACCEPTS_LAZY_VOXELS = True       # forward() reads the voxel count back together with its table plan's level sizes

    def _finish(self, batch_dict, out, feats):
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': dict(zip(('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4'), feats))})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict

    def _chain(self):
        return [self.conv_input, self.conv1, self.conv2, self.conv3, self.conv4, self.conv_out]

    def _input_tensor(self, batch_dict):
        voxel_features, voxel_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        return spconv.SparseConvTensor(features=voxel_features, indices=voxel_coords.int().contiguous(),
                                       spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])

    def prefetch(self, batch_dict):
        """The part of the table plan that needs no host decision, for a LATER forward(batch_dict) on the same stream (the detector's
        prefetch_sparse: called while the previous batch's backward pass is still to be enqueued): the strided levels of the lazily
        voxelized batch are marked and counted, the counts go to pinned host memory behind an event. forward() then finds them
        there instead of stalling the host - and the launch queue behind it - on a read-back in the middle of the sparse phase.
        No-op for batches that are not lazily voxelized."""
        n_dev = batch_dict.get('voxel_count_dev', None)
        if n_dev is None or not PLAN_INDICES or '_sparse_prefetch' in batch_dict:
            return batch_dict
        from spconv.pytorch.conv import plan_begin
        x = self._input_tensor(batch_dict)
        pending = plan_begin(self._chain(), x, n_dev)
        if pending is not None:
            batch_dict['_sparse_prefetch'] = (x, pending)
        return batch_dict

    def forward(self, batch_dict):
        n_dev = batch_dict.get('voxel_count_dev', None)          # lazily voxelized batch: capacity buffers, row count on the device
        if n_dev is not None and not PLAN_INDICES:
            from .vfe.mean_vfe import finish_lazy_voxels
            finish_lazy_voxels(batch_dict)
            n_dev = None
        x, pending = batch_dict.pop('_sparse_prefetch', (None, None))
        if x is None or n_dev is None:
            x, pending = self._input_tensor(batch_dict), None
        if PLAN_INDICES:
            # all rulebooks first: one host read-back for the four strided output sets instead of a sync per strided layer (and
            # for the voxel count of a lazily voxelized batch: x is cut to its rows by the plan); a prefetched batch has its
            # counts on the host already
            from crbhip import bnrelu
            plan_indices(self._chain(), x, with_frame_offsets=bnrelu.active_groups() is not None, n_dev=n_dev, pending=pending)
            if n_dev is not None:
                from .vfe.mean_vfe import finish_lazy_voxels
                finish_lazy_voxels(batch_dict, x.indices.shape[0])
        if torch.is_grad_enabled() and x.features.is_cuda:
            # forward and input-gradient weight layouts of every sparse layer of this step in one launch
            from crbhip import sparse as _sp
            if _sp.PREPARE_WEIGHTS:
                _sp.prepare_weights([m for m in self.modules() if isinstance(m, spconv.SparseConvolution) and not m.conv1x1])
        x = self.conv_input(x)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        out = self.conv_out(x4)
        if torch.is_grad_enabled() and x.features.is_cuda:
            from crbhip import sparse as _sp
            _sp._PREP_W.clear()          # the forward operands were for THIS pass (the backward finds its operands by their own address)
        return self._finish(batch_dict, out, (x1, x2, x3, x4))


class VoxelBackBone8x(_Backbone8xBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(grid_size[2]) + 1, int(grid_size[1]), int(grid_size[0])]
        block = post_act_block
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(   # [1600,1408,41] -> [800,704,21]
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(   # -> [400,352,11]
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(   # -> [200,176,5]
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(  # -> [200,176,2]
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}


class VoxelResBackBone8x(_Backbone8xBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(grid_size[2]) + 1, int(grid_size[1]), int(grid_size[0])]
        block = post_act_block
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        self.conv1 = spconv.SparseSequential(
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'))
        last_pad = self.model_cfg.get('last_pad', 0) if hasattr(self.model_cfg, 'get') else 0
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                                indice_key='spconv_down2'), norm_fn(128), nn.ReLU())
        self.num_point_features = 128
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 128}
