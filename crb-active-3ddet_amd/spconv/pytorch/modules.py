"""SparseModule / SparseSequential with the container behaviour the reference relies on
(pcdet/models/backbones_3d/spconv_backbone.py:21-25,77-117)."""
from collections import OrderedDict

import torch
from torch import nn

from crbhip import bnrelu as _bnrelu
from .core import SparseConvTensor

FUSE_CONV_BN_EVAL = True  # inference (no grad, BatchNorm1d in eval mode): conv -> BatchNorm1d -> ReLU triples run as ONE launch,
                          # normalisation and ReLU applied to the accumulator (crb_sparse_conv_forward_compact_bn)
FUSE_BN_RELU = True      # run BatchNorm1d -> ReLU pairs that follow a sparse conv as one fused HIP op (same modules,
                         # same parameters / buffers / state_dict; set False for the plain torch path)


class SparseModule(nn.Module):
    """marker base class: modules that take and return a SparseConvTensor"""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError('name exists')
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if (FUSE_CONV_BN_EVAL and not torch.is_grad_enabled() and hasattr(module, 'fuses_bn_eval') and
                        isinstance(nxt, nn.BatchNorm1d) and not nxt.training and nxt.affine and
                        nxt.running_mean is not None and module.fuses_bn_eval(input.features)):
                    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                    input = module(input, epilogue=(nxt, relu))
                    i += 1 + int(relu)                  # BatchNorm1d (and ReLU) consumed by the conv's epilogue
                else:
                    input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    feats = input.features
                    if (FUSE_BN_RELU and isinstance(module, nn.BatchNorm1d) and i + 1 < len(mods) and
                            isinstance(mods[i + 1], nn.ReLU) and _bnrelu.supported(feats, module)):
                        grp = _bnrelu.active_groups()
                        if grp is not None and module.training:
                            grp.note_rows(feats, input.frame_offsets)        # sparse rows are ragged per frame
                        input = input.replace_feature(_bnrelu.bn_relu(feats, module, relu=True))
                        i += 1                      # the ReLU module was consumed by the fused op
                    else:
                        input = input.replace_feature(module(feats))
            else:
                input = module(input)
            i += 1
        return input
