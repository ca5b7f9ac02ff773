"""BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:6-112): dense 2-D convs. Same module tree => same
state_dict keys. The stride-1 3x3 convolutions run on the hand-written Winograd kernel (below), the rest on MIOpen."""
import numpy as np
import torch
import torch.nn as nn

from crbhip import bnrelu

from ...utils.fold_utils import fold_conv_bn
from ...utils.linear_rows import LinearRows, rows_view, rows_to_nchw


ROWS_TRAIN = True      # BatchNorm2d+ReLU pairs through the fused row kernels when the activations are channels_last
ROWS_GEMM = True       # kernel-1 stride-1 up-sampling branch (ConvTranspose2d 1x1) as a row GEMM on the channels_last map
# The stride-1 3x3 convolutions (11 of the 12 of the KITTI config) run as hand-written F(2x2,3x3) Winograd on the f32 MFMA
# (crbhip.winograd, csrc/winograd_conv2.hip) instead of MIOpen's implicit GEMM: forward (in eval with BatchNorm folded into the
# transformed weights and bias + ReLU in the kernel's epilogue: one launch per layer) and input gradient; the weight gradient
# stays on MIOpen. Results equal the direct convolution up to f32 rounding of the transforms (4e-7 of the output scale against an
# f64 convolution; MIOpen's own error there is 1.2e-6). Round 4: 0.77 ms per 128->128 @ 16x200x176 call against MIOpen's 1.41 —
# the default; CRB_WINOGRAD=0 (or the flag) gives the MIOpen path back.
WINOGRAD = __import__('os').environ.get('CRB_WINOGRAD', '1') != '0'


def _wino_fold(w, shift):
    """fold_conv_bn transform: folded conv weight (Cout,Cin,3,3) -> weight image of the Winograd kernel, bias"""
    from crbhip import winograd
    return winograd.weights_forward2(w), shift.contiguous()


def _wino_ok(conv, x, pad=None):
    if not WINOGRAD or not isinstance(conv, nn.Conv2d) or not x.is_cuda or x.dtype != torch.float32:
        return False
    from crbhip import winograd
    p = tuple(conv.padding) if pad is None else tuple(pad)
    return conv.kernel_size == (3, 3) and conv.stride == (1, 1) and p == (1, 1) and conv.dilation == (1, 1) and \
        conv.groups == 1 and conv.padding_mode == 'zeros' and \
        winograd.supported2(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3]) and \
        x.is_contiguous(memory_format=torch.channels_last)


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


class BaseBEVBackbone(nn.Module):
    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(model_cfg.get('LAYER_NUMS', None) or [])
        layer_strides = list(model_cfg.get('LAYER_STRIDES', None) or [])
        num_filters = list(model_cfg.get('NUM_FILTERS', None) or [])
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        upsample_strides = list(model_cfg.get('UPSAMPLE_STRIDES', None) or [])
        num_upsample_filters = list(model_cfg.get('NUM_UPSAMPLE_FILTERS', None) or [])
        assert len(upsample_strides) == len(num_upsample_filters)
        c_in_list = [input_channels] + num_filters[:-1]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx, (c_in, c_out, n, s) in enumerate(zip(c_in_list, num_filters, layer_nums, layer_strides)):
            layers = [nn.ZeroPad2d(1), nn.Conv2d(c_in, c_out, kernel_size=3, stride=s, padding=0, bias=False),
                      _bn(c_out), nn.ReLU()]
            for _ in range(n):
                layers += [nn.Conv2d(c_out, c_out, kernel_size=3, padding=1, bias=False), _bn(c_out), nn.ReLU()]
            self.blocks.append(nn.Sequential(*layers))
            if upsample_strides:
                us, uc = upsample_strides[idx], num_upsample_filters[idx]
                if us >= 1:
                    up = nn.ConvTranspose2d(c_out, uc, us, stride=us, bias=False)
                else:
                    ds = int(np.round(1 / us))
                    up = nn.Conv2d(c_out, uc, ds, stride=ds, bias=False)
                self.deblocks.append(nn.Sequential(up, _bn(uc), nn.ReLU()))
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > len(layer_nums):
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                _bn(c_in), nn.ReLU()))
        self.num_bev_features = c_in

    @staticmethod
    def _run_folded(seq, x):
        """inference only: every (Conv2d | ConvTranspose2d) -> BatchNorm2d -> ReLU triple runs as the convolution followed by
        ONE elementwise pass. channels_last CUDA tensors: the conv output viewed as (N*H*W, C) rows goes through
        crb_bn_relu_apply (scale/shift from the running statistics + ReLU, in place) — with the BN folded into the conv
        weights PyTorch still launches a separate bias-add and a separate ReLU pass over the 288 MB BEV tensors. Other
        tensors: BN folded into the weights (same values up to f32 rounding)."""
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            pad = None                                   # ZeroPad2d(p) + Conv2d(padding=0) -> Conv2d(padding=p), no padded copy
            if isinstance(m, nn.ZeroPad2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Conv2d):
                c, pd = mods[i + 1], m.padding
                if c.padding == (0, 0) and c.padding_mode == 'zeros' and pd[0] == pd[1] and pd[2] == pd[3]:
                    pad, i, m = (pd[2], pd[0]), i + 1, c
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
                bn = mods[i + 1]
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                if _wino_ok(m, x, pad):
                    from crbhip import winograd
                    U, shift = fold_conv_bn(m, bn, _wino_fold)
                    x = winograd.conv3x3_U2(x, U, shift, relu)
                    i += 3 if relu else 2
                    continue
                rows_ok = x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and m.bias is None
                if rows_ok:
                    y = m(x) if pad is None else torch.nn.functional.conv2d(x, m.weight, None, m.stride, pad, m.dilation,
                                                                             m.groups)
                    rows_ok = y.is_contiguous(memory_format=torch.channels_last) and \
                        bnrelu.supported(y.new_empty((2, y.shape[1])), bn)
                if rows_ok:
                    n, c, h, w_ = y.shape
                    rows = y.permute(0, 2, 3, 1).reshape(n * h * w_, c)          # a view of the NHWC storage
                    z = bnrelu.bn_apply_(rows, bn, relu)
                    x = z.view(n, h, w_, c).permute(0, 3, 1, 2)
                else:
                    w, shift = fold_conv_bn(m, bn)
                    if isinstance(m, nn.Conv2d):
                        x = torch.nn.functional.conv2d(x, w, shift, m.stride, m.padding if pad is None else pad, m.dilation,
                                                       m.groups)
                    else:
                        x = torch.nn.functional.conv_transpose2d(x, w, shift, m.stride, m.padding, m.output_padding,
                                                                 m.groups, m.dilation)
                    if relu:
                        x = torch.relu_(x)
                i += 3 if relu else 2
            else:
                x = m(x) if pad is None else torch.nn.functional.conv2d(x, m.weight, m.bias, m.stride, pad, m.dilation, m.groups)
                i += 1
        return x

    @staticmethod
    def _pad_conv(mods, i, x):
        """ZeroPad2d(p) followed by Conv2d(padding=0) == the same conv with padding=p: the explicit pad makes a padded copy of
        the (B,256,200,176) map (and a slice copy in backward). -> (output, modules consumed) or (None, 0)"""
        m = mods[i]
        if isinstance(m, nn.ZeroPad2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.Conv2d):
            c = mods[i + 1]
            pd = m.padding
            if c.padding == (0, 0) and c.padding_mode == 'zeros' and pd[0] == pd[1] and pd[2] == pd[3]:
                if _wino_ok(c, x, (pd[2], pd[0])):
                    from crbhip import winograd
                    return BaseBEVBackbone._wino_conv(c, x, mods[i + 2] if i + 2 < len(mods) else None), 2
                if torch.are_deterministic_algorithms_enabled() and torch.is_grad_enabled() and pd[0] == pd[2] and x.is_cuda:
                    from crbhip import dense_strided       # (MIOpen's weight gradient of this layer adds split-K partials with atomics)
                    if dense_strided.supported(c, x):
                        return dense_strided.conv_det(c, x, (pd[2], pd[0])), 2
                return torch.nn.functional.conv2d(x, c.weight, c.bias, c.stride, (pd[2], pd[0]), c.dilation, c.groups), 2
        return None, 0

    @staticmethod
    def _wino_conv(conv, x, nxt):
        """stride-1 3x3 convolution on the Winograd kernels; when a training-mode BatchNorm2d follows (and the conv has no bias) the
        forward kernel also writes the slab sums of its output (crb_conv3x3_winograd2_stats_nhwc), attached to the result for
        _run_rows_train: that BatchNorm launches no statistics pass"""
        from crbhip import winograd
        if winograd.STATS and conv.bias is None and isinstance(nxt, nn.BatchNorm2d) and nxt.training and \
                nxt.momentum is not None and bnrelu.FUSE_RUNNING and torch.is_grad_enabled() and not bnrelu.frame_groups_active():
            y, slabs = winograd.conv3x3_stats(x, conv.weight)
            y._crb_bn_slabs = slabs
            return y
        return winograd.conv3x3(x, conv.weight, conv.bias)

    @staticmethod
    def _run_rows_train(seq, x):
        """training / grad-enabled path on channels_last CUDA tensors: the BatchNorm2d -> ReLU pairs run as the fused row
        kernels on the (N*H*W, C) view of the NHWC storage (crb_bn_relu_forward / _backward: statistics pass + apply pass
        forward, reduce + apply pass backward with the ReLU mask recomputed) instead of MIOpen BatchNorm2d plus separate
        ReLU forward / backward passes over the 288 MB BEV tensors."""
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            y, used = BaseBEVBackbone._pad_conv(mods, i, x)
            if used:
                x, i = y, i + used
                continue
            if _wino_ok(m, x):
                x = BaseBEVBackbone._wino_conv(m, x, nxt)
                i += 1
                continue
            if isinstance(m, nn.BatchNorm2d) and isinstance(nxt, nn.ReLU) and \
                    x.is_contiguous(memory_format=torch.channels_last) and \
                    bnrelu.supported(x.new_empty((2, x.shape[1])), m):
                n, c, h, w_ = x.shape
                rows = x.permute(0, 2, 3, 1).reshape(n * h * w_, c)
                slabs = getattr(x, '_crb_bn_slabs', None)           # written by the Winograd forward kernel that produced x
                x = bnrelu.bn_relu(rows, m, relu=True, slabs=slabs).view(n, h, w_, c).permute(0, 3, 1, 2)
                i += 2
            else:
                x = m(x)
                i += 1
        return x

    @staticmethod
    def _up(conv, x):
        """first module of an up-sampling branch. A kernel-1 stride-1 ConvTranspose2d / Conv2d without bias on a
        channels_last CUDA map is the GEMM rows @ W (no convolution kernel); everything else runs as the module."""
        if ROWS_GEMM and x.is_cuda and conv.bias is None and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and \
                conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1):
            rows = rows_view(x)
            if rows is not None:
                w2d = conv.weight[:, :, 0, 0]
                w2d = w2d.t() if isinstance(conv, nn.ConvTranspose2d) else w2d            # (Cout, Cin)
                n, _, h, w_ = x.shape
                return rows_to_nchw(LinearRows.apply(rows, w2d), n, h, w_)
        if torch.are_deterministic_algorithms_enabled() and torch.is_grad_enabled() and x.is_cuda and isinstance(conv, nn.ConvTranspose2d):
            from crbhip import dense_strided
            if dense_strided.supported(conv, x):
                return dense_strided.conv_det(conv, x)
        return conv(x)

    def _can_fuse_concat_eval(self, x):
        if len(self.deblocks) != len(self.blocks) or len(self.deblocks) < 2:
            return False
        for d in self.deblocks:
            m = list(d)
            if len(m) != 3 or not isinstance(m[1], nn.BatchNorm2d) or not isinstance(m[2], nn.ReLU) or m[0].bias is not None \
                    or not bnrelu.supported(x.new_empty((2, m[1].num_features)), m[1]):
                return False
        return True

    def _can_fuse_concat(self, x):
        """training + channels_last + every up-sampling branch is [conv, BatchNorm2d(train, momentum), ReLU] with a channel
        count the row kernels take, and there is no extra deblock after the concat"""
        if not self.training or len(self.deblocks) != len(self.blocks) or len(self.deblocks) < 2:
            return False
        for d in self.deblocks:
            m = list(d)
            if len(m) != 3 or not isinstance(m[1], nn.BatchNorm2d) or not isinstance(m[2], nn.ReLU) or \
                    not m[1].training or m[1].momentum is None or \
                    not bnrelu.supported(x.new_empty((2, m[1].num_features)), m[1]):
                return False
        return x.is_contiguous(memory_format=torch.channels_last)

    def forward(self, data_dict):
        spatial_features = data_dict['spatial_features']
        ups = []
        x = spatial_features
        if not self.training and not torch.is_grad_enabled():
            cat_ok = x.is_cuda and x.is_contiguous(memory_format=torch.channels_last) and self._can_fuse_concat_eval(x)
            pre = []
            for i, blk in enumerate(self.blocks):
                x = self._run_folded(blk, x)
                if cat_ok:
                    pre.append(self._up(self.deblocks[i][0], x))
                else:
                    ups.append(self._run_folded(self.deblocks[i], x) if len(self.deblocks) > 0 else x)
            if cat_ok and all(p.is_contiguous(memory_format=torch.channels_last) and p.shape[2:] == pre[0].shape[2:]
                              for p in pre):
                # the up-sampled branches' BN+ReLU write their channel slices of the concatenated map directly
                n, _, h, w_ = pre[0].shape
                cat = torch.empty((n * h * w_, sum(p.shape[1] for p in pre)), dtype=torch.float32, device=x.device)
                col = 0
                for d, p in zip(self.deblocks, pre):
                    bnrelu.bn_apply_into(p.permute(0, 2, 3, 1).reshape(n * h * w_, p.shape[1]), d[1], True, cat, col)
                    col += p.shape[1]
                ups = [cat.view(n, h, w_, cat.shape[1]).permute(0, 3, 1, 2)]
            elif cat_ok:
                ups = [self._run_folded(nn.Sequential(*list(d)[1:]), p) for d, p in zip(self.deblocks, pre)]
            x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
            if len(self.deblocks) > len(self.blocks):
                x = self._run_folded(self.deblocks[-1], x)
            data_dict['spatial_features_2d'] = x
            return data_dict
        run = self._run_rows_train if (ROWS_TRAIN and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last)) \
            else (lambda seq, t: seq(t))
        if run == self._run_rows_train and torch.is_grad_enabled():
            # the Winograd weight images of all stride-1 3x3 layers of this step (forward + input gradient) in one launch
            from crbhip import winograd
            if winograd.PREPARE:
                winograd.prepare_weights2([m.weight for blk in self.blocks for m in blk
                                           if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1)
                                           and m.weight.is_cuda])
        fuse_cat = ROWS_TRAIN and x.is_cuda and self._can_fuse_concat(x)
        pre = []                                         # deblock conv outputs awaiting their joint BN+ReLU+concat
        for i, blk in enumerate(self.blocks):
            x = run(blk, x)
            stride = int(spatial_features.shape[2] / x.shape[2])
            data_dict['spatial_features_%dx' % stride] = x
            if fuse_cat:
                pre.append(self._up(self.deblocks[i][0], x))       # ConvTranspose2d / Conv2d only
            else:
                ups.append(run(self.deblocks[i], x) if len(self.deblocks) > 0 else x)
        if fuse_cat and all(p.is_contiguous(memory_format=torch.channels_last) and p.shape[2:] == pre[0].shape[2:]
                            for p in pre):
            n, _, h, w_ = pre[0].shape
            rows = [p.permute(0, 2, 3, 1).reshape(n * h * w_, p.shape[1]) for p in pre]
            cat = bnrelu.bn_relu_concat(rows, [d[1] for d in self.deblocks[:len(pre)]], relu=True)
            ups = [cat.view(n, h, w_, cat.shape[1]).permute(0, 3, 1, 2)]
        elif fuse_cat:
            ups = [d[2](d[1](p)) for d, p in zip(self.deblocks, pre)]
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > len(self.blocks):
            x = self.deblocks[-1](x)
        data_dict['spatial_features_2d'] = x
        if x.is_cuda:
            from crbhip import winograd
            winograd.forget_prepared_forward()
        return data_dict
