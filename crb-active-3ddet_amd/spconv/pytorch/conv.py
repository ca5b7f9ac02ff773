"""Sparse convolution layers (host mirror of spconv.pytorch.conv) over the gfx950 gather-GEMM kernels.

Parameter layout follows spconv 2.x: weight (Cout, kd, kh, kw, Cin) — the layout the reference's checkpoint
loader adapts to (pcdet/models/detectors/detector3d_template.py:455-484)."""
import math

import torch
from torch import nn
from torch.nn import init

from crbhip import sparse as _sp
from crbhip import bnrelu as _bnrelu
from .core import SparseConvTensor
from .modules import SparseModule


def _ntuple(v, n):
    if isinstance(v, (list, tuple)):
        assert len(v) == n
        return [int(x) for x in v]
    return [int(v)] * n


class SparseConvolution(SparseModule):
    # arithmetic contract of this layer's gather-GEMM (forward + dgrad): 'f32' exact (default) or the opt-in 'bf16x3'
    # (crbhip.sparse); per module, set with spconv.pytorch.set_arithmetic(model, ...) — not process state
    arithmetic = 'f32'

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 algo=None, fp32_accum=None, name=None):
        super().__init__()
        assert groups == 1, 'groups > 1 is not on the reference hot path'
        assert ndim in (2, 3)
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _ntuple(kernel_size, ndim)
        self.stride = _ntuple(stride, ndim)
        self.padding = _ntuple(padding, ndim)
        self.dilation = _ntuple(dilation, ndim)
        assert all(d == 1 for d in self.dilation), 'dilation != 1 is not on the reference hot path'
        assert not transposed, 'SparseConvTranspose is not on the reference hot path'
        self.subm, self.inverse, self.transposed = subm, inverse, transposed
        self.indice_key = indice_key
        self.conv1x1 = all(k == 1 for k in self.kernel_size) and all(s == 1 for s in self.stride)
        self.weight = nn.Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def extra_repr(self):
        return (f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, '
                f'padding={self.padding}, subm={self.subm}, inverse={self.inverse}, indice_key={self.indice_key}')

    def reset_parameters(self):
        # kaiming-uniform like torch.nn.Conv* / spconv: fan_in = Cin * prod(k)
        fan_in = self.in_channels
        for k in self.kernel_size:
            fan_in *= k
        gain = math.sqrt(2.0 / (1 + 5.0))           # a = sqrt(5)
        bound = gain * math.sqrt(3.0 / fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                b = 1.0 / math.sqrt(fan_in)
                init.uniform_(self.bias, -b, b)

    def _k3(self, v, fill):
        return ([fill] + list(v)) if self.ndim == 2 else list(v)

    def fuses_bn_eval(self, feats):
        """can this conv take the following BatchNorm1d(eval) + ReLU as an epilogue of its forward kernel?"""
        return (not self.inverse and not self.conv1x1 and feats.is_cuda and feats.shape[0] > 0 and
                self.arithmetic == 'f32' and _sp.epilogue_supported(self.in_channels, self.out_channels))

    def train(self, mode=True):
        # the inference layout cache (weight_kio) is keyed on the parameter's address and autograd version; writes through `.data`
        # (the reference OptimWrapper's weight decay, an fp16 master copy, EMA updates) bump neither: the cache does not outlive a
        # switch between training and evaluation, nor a load_state_dict
        self.__dict__.pop('_crb_kio', None)
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop('_crb_kio', None)
        return super()._load_from_state_dict(*args, **kwargs)

    def weight_kio(self):
        """(Cout,k..,Cin) -> (K,Cin,Cout) contiguous; differentiable. Inference keeps the layout on the module until the parameter's
        address / autograd version changes or the module changes mode; in-place writes through `.data` while the module STAYS in
        eval mode are not detected (call module.train(False) again, or drop `_crb_kio`)."""
        w = self.weight
        K = 1
        for k in self.kernel_size:
            K *= k
        if w.is_cuda and not torch.is_grad_enabled():
            # inference: the layout is a function of the parameter alone - kept on the module until the parameter changes (address /
            # autograd version, as fold_conv_bn keys its folded weights): no permuted copy per layer per batch of a scoring pass
            key = (w.data_ptr(), w._version, tuple(w.shape))
            hit = self.__dict__.get('_crb_kio')
            if hit is None or hit[0] != key:
                hit = self.__dict__['_crb_kio'] = (key, w.detach().reshape(self.out_channels, K, self.in_channels).permute(1, 2, 0)
                                                   .contiguous())
            return hit[1]
        if w.is_cuda:
            return _sp.weight_kio(w, K)          # (the copy made ahead for all layers of the step, when there is one)
        return w.reshape(self.out_channels, K, self.in_channels).permute(1, 2, 0).contiguous()

    def forward(self, input, epilogue=None):
        """epilogue = (BatchNorm1d in eval mode, relu flag): inference only (SparseSequential decides), the conv bias, the
        normalisation and the ReLU are then applied inside the forward kernel"""
        assert isinstance(input, SparseConvTensor)
        feats, indices = input.features, input.indices
        ndim = self.ndim
        spatial_shape = input.spatial_shape
        if ndim == 2:
            idx3 = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1).contiguous()
            shape3 = [1] + list(spatial_shape)
        else:
            idx3, shape3 = indices, list(spatial_shape)
        ks, st, pd = self._k3(self.kernel_size, 1), self._k3(self.stride, 1), self._k3(self.padding, 0)

        if self.conv1x1 and not self.inverse:
            out = torch.mm(feats, self.weight.view(self.out_channels, self.in_channels).t())
            if self.bias is not None:
                out = out + self.bias
            return input.replace_feature(out)

        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None, 'inverse conv needs the indice_key of a SparseConv'
            rb = datas
            assert not rb.subm
            out_feats = _sp.sparse_conv(feats, self.weight_kio(), rb, True, self.arithmetic)
            out_indices3, out_shape3 = rb.in_coords, rb.in_shape
        else:
            if datas is not None and self.subm:
                rb = datas
                assert rb.n_in == feats.shape[0], 'indice_key reused on a tensor with a different active set'
            elif datas is not None and not self.subm:
                rb = datas
                assert rb.n_in == feats.shape[0]
            else:
                if self.subm:
                    rb = _sp.subm_rulebook(idx3, shape3, ks)
                else:
                    rb = _sp.spconv_rulebook(idx3, shape3, input.batch_size, ks, st, pd)
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = rb
            if epilogue is not None:
                out_feats = _sp.sparse_conv_bn_eval(feats, self.weight_kio(), rb, self.bias, epilogue[0], epilogue[1])
            else:
                out_feats = _sp.sparse_conv(feats, self.weight_kio(), rb, False, self.arithmetic)
            out_indices3, out_shape3 = (idx3, shape3) if self.subm else (rb.out_coords, rb.out_shape)
        if self.bias is not None and epilogue is None:
            out_feats = out_feats + self.bias
        if ndim == 2:
            out_indices = out_indices3[:, [0, 2, 3]].contiguous()
            out_shape = out_shape3[1:]
        else:
            out_indices, out_shape = out_indices3, out_shape3
        out = SparseConvTensor(out_feats, out_indices, out_shape, input.batch_size, input.grid, input.voxel_num,
                               input.indice_dict, input.benchmark)
        if self.inverse:
            out.frame_offsets = getattr(rb, 'in_frame_offsets', None)
        elif self.subm:
            out.frame_offsets = input.frame_offsets
        else:
            out.frame_offsets = getattr(rb, 'out_frame_offsets', None)
        grp = _bnrelu.active_groups()
        if grp is not None and out.frame_offsets is not None:
            grp.note_rows(out_feats, out.frame_offsets)             # a BatchNorm called on these rows finds their frames
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True, algo=None, fp32_accum=None,
                 name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, True,
                         indice_key=indice_key)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


def set_arithmetic(module, arithmetic):
    """opt a model (or one layer) in to / out of the 'bf16x3' gather-GEMM contract: sets .arithmetic on every
    SparseConvolution below `module`; returns the number of layers touched"""
    if arithmetic not in _sp.ARITHMETICS:
        raise ValueError('arithmetic must be one of %s' % (_sp.ARITHMETICS,))
    if arithmetic == 'bf16x3':
        _sp._bf16x3_supported(64, 64)            # (measurement library only since round 5: raises in a product process)
    n = 0
    for m in module.modules():
        if isinstance(m, SparseConvolution):
            m.arithmetic = arithmetic
            n += 1
    return n


def _chain_specs(modules, input):
    """the convs of `modules` that need a rulebook, in forward order, and the chain description build_rulebooks takes
    -> (convs, have flags of the strided keys, specs, keys to fill)"""
    convs = []
    for root in modules:
        for m in root.modules():
            if isinstance(m, SparseConvolution) and m.ndim == 3 and not m.inverse and not m.conv1x1 and \
                    m.indice_key is not None:
                convs.append(m)
    have = [m.indice_key in input.indice_dict for m in convs if not m.subm]
    specs, todo, seen = [], [], set(input.indice_dict.keys())
    shape = list(input.spatial_shape)
    geom_of = {}
    for m in convs:
        key = m.indice_key
        ks = m._k3(m.kernel_size, 1)
        if m.subm:
            g = ('subm', tuple(ks), tuple(shape))
        else:
            g = ('spconv', tuple(ks), tuple(m._k3(m.stride, 1)), tuple(m._k3(m.padding, 0)), tuple(shape))
        if key in geom_of:
            if geom_of[key] != g:
                raise ValueError('indice_key %r is shared by convs of different geometry or level: %s vs %s' % (key, geom_of[key], g))
        else:
            geom_of[key] = g
            if key not in seen:
                specs.append(g[:2] if m.subm else g[:4])
                todo.append(key)
        if not m.subm:
            shape = _sp.conv_out_shape(shape, g[1], g[2], g[3])
    return convs, have, specs, todo


def plan_begin(modules, input, n_dev):
    """The device-only first half of plan_indices(..., n_dev=...) for a lazily voxelized tensor, issued ahead of time with no host
    synchronisation (crbhip.sparse.begin_rulebooks): the strided levels are marked and counted, the counts travel to pinned host
    memory behind an event. -> PendingChain to pass as plan_indices(..., pending=...) before the forward pass, or None when the
    tensor is not a fresh unplanned chain (plan_indices then does everything itself)."""
    convs, have, specs, todo = _chain_specs(modules, input)
    if any(have) or not specs or not any(s[0] == 'spconv' for s in specs):
        return None
    return _sp.begin_rulebooks(input.indices, list(input.spatial_shape), input.batch_size, specs, n_dev)


def plan_indices(modules, input, with_frame_offsets=False, n_dev=None, pending=None):
    """Build the rulebooks of every (non 1x1, non inverse) SparseConvolution found in `modules` (in module order, as the
    forward pass will meet them) BEFORE the first feature kernel runs, and leave them in input.indice_dict under their
    indice_key: crbhip.sparse.build_rulebooks takes the whole chain — the output sets of all strided layers are marked and
    counted with ONE host read-back, every table costs one launch and all of them are finished (kernel order, compact
    tables, tile order, wgrad pair lists) by two more — instead of ~30 launches and a synchronisation per layer in the
    middle of the forward pass.
    Precondition (checked): the convs that carry a new indice_key form one linear chain starting at `input` — every strided
    conv consumes the output set of the strided conv before it. A tensor on which some of the strided keys already exist
    is left alone: the forward pass then builds what is missing layer by layer.
    n_dev (1,) i32 cuda: input.features / input.indices are the voxel generator's capacity buffers and the row count is still on
    the device; the chain's read-back returns it and the tensor is cut to its rows here (input.indices.shape[0] afterwards)."""
    convs, have, specs, todo = _chain_specs(modules, input)

    def cut(n):
        input.indices = input.indices[:n]
        input.features = input.features[:n]
        return n
    if any(have) and not all(have):
        # partially planned tensor: no assumption about where the chain stands
        if n_dev is not None:
            cut(int(n_dev.cpu()[0]))
        return input
    n_rows = None
    if specs and not (have and all(have)):
        books = _sp.build_rulebooks(input.indices, list(input.spatial_shape), input.batch_size, specs,
                                    want_grad=torch.is_grad_enabled(), n_dev=n_dev, pending=pending)
        if n_dev is not None or pending is not None:
            books, n_rows = books
            cut(n_rows)
        for key, rb in zip(todo, books):
            input.indice_dict[key] = rb
    elif n_dev is not None:
        n_rows = cut(int(n_dev.cpu()[0]))
    # (all strided keys present, SubM keys missing: each is built on its level by the forward pass)
    if with_frame_offsets:
        # rows per frame of the input and of every strided level (sparse rows are frame-sorted): ONE read-back. Used by the
        # per-frame BatchNorm of batched CRB stage 2 (crbhip.bnrelu.frame_groups).
        B = input.batch_size
        levels = [('in', input.indices)] + [(m.indice_key, input.indice_dict[m.indice_key].out_coords)
                                            for m in convs if not m.subm]
        counts = torch.zeros((len(levels), B), dtype=torch.int64, device=input.indices.device)
        for k, (_, c) in enumerate(levels):
            if c.shape[0]:
                counts[k].scatter_add_(0, c[:, 0].long(), torch.ones((c.shape[0],), dtype=torch.int64, device=c.device))
        host = counts.cpu().tolist()
        offs = [[0] + [sum(row[:k + 1]) for k in range(B)] for row in host]
        input.frame_offsets = offs[0]
        prev = offs[0]
        for (key, _), o in zip(levels[1:], offs[1:]):
            input.indice_dict[key].in_frame_offsets = prev
            input.indice_dict[key].out_frame_offsets = o
            prev = o
    return input
