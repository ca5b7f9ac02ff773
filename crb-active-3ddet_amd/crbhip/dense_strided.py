"""Deterministic mode for the two strided dense convolutions of the BEV backbone (pcdet/models/backbones_2d/base_bev_backbone.py:33-37
the stride-2 3x3 block entry, :50-55 the kernel-2 stride-2 ConvTranspose2d up-sampling branch).

Their forward and data gradient stay with the vendor library (MIOpen / CK: run-to-run identical here); MIOpen's WEIGHT gradient of
both is a split-K implicit GEMM that adds its partials with float atomics (`igemm_wrw_..._gkgs`): the two parameters were the only
ones of a SECOND training step whose gradients differed between runs (tools/dbg_determinism.py). With
torch.are_deterministic_algorithms_enabled() the weight gradient is computed by the gather-GEMM weight-gradient kernel of the sparse
backbone instead (crb_sparse_conv_wgrad: partials reduced by a fixed-shape tree) on a dense rulebook: a channels_last map IS a row
matrix, a strided convolution a regular pair list. Slower than MIOpen's kernel (the pair lists of all taps are walked, channel blocks
above 128 are copied out), which is why it is a mode and not the default."""
import torch

from . import sparse

_RULEBOOKS = {}
_BLOCK = 128


def _pairs(kind, N, H, W, kh, kw, s, p, dev):
    """(pair_in, pair_out, pair_start) of the taps (a, b) in row-major order; 'in' = pixels of the convolution's INPUT map, 'out' =
    pixels of its output map, pair_out ascending inside a tap. kind 'conv': out (i, j) reads in (i s + a - p, j s + b - p);
    kind 'deconv' (ConvTranspose2d, padding 0): in (i, j) writes out (i s + a, j s + b)."""
    key = (kind, N, H, W, kh, kw, s, p, dev.index)
    rb = _RULEBOOKS.get(key)
    if rb is not None:
        return rb
    n = torch.arange(N, device=dev, dtype=torch.int64).view(N, 1, 1)
    pin, pout, start = [], [], [0]
    if kind == 'conv':
        Ho, Wo = (H + 2 * p - kh) // s + 1, (W + 2 * p - kw) // s + 1
        i = torch.arange(Ho, device=dev, dtype=torch.int64).view(1, Ho, 1)
        j = torch.arange(Wo, device=dev, dtype=torch.int64).view(1, 1, Wo)
        for a in range(kh):
            for b in range(kw):
                u, v = i * s + a - p, j * s + b - p
                ok = (((u >= 0) & (u < H)) & ((v >= 0) & (v < W))).expand(N, Ho, Wo)
                pin.append(((n * H + u) * W + v).expand(N, Ho, Wo)[ok])
                pout.append(((n * Ho + i) * Wo + j).expand(N, Ho, Wo)[ok])
                start.append(start[-1] + int(pin[-1].numel()))
    else:
        Hy, Wy = (H - 1) * s + kh, (W - 1) * s + kw
        i = torch.arange(H, device=dev, dtype=torch.int64).view(1, H, 1)
        j = torch.arange(W, device=dev, dtype=torch.int64).view(1, 1, W)
        for a in range(kh):
            for b in range(kw):
                pin.append(((n * H + i) * W + j).expand(N, H, W).reshape(-1))
                pout.append(((n * Hy + i * s + a) * Wy + j * s + b).expand(N, H, W).reshape(-1))
                start.append(start[-1] + int(pin[-1].numel()))
    rb = (torch.cat(pin).to(torch.int32), torch.cat(pout).to(torch.int32), torch.tensor(start, dtype=torch.int32, device=dev))
    if len(_RULEBOOKS) > 16:
        _RULEBOOKS.clear()
    _RULEBOOKS[key] = rb
    return rb


def weight_grad(kind, x, dy, kh, kw, s, p):
    """dW of y = conv2d(x, w, stride s, padding p) (kind 'conv': (Cout, Cin, kh, kw)) or y = conv_transpose2d(x, w, stride s)
    (kind 'deconv': (Cin, Cout, kh, kw)) from channels_last x (N, Cin, H, W) and dy, without atomics."""
    N, cin, H, W = x.shape
    cout = dy.shape[1]
    xr = x.permute(0, 2, 3, 1).reshape(N * H * W, cin)
    dyr = dy.permute(0, 2, 3, 1).reshape(-1, cout)
    pairs = _pairs(kind, N, H, W, kh, kw, s, p, x.device)
    K = kh * kw
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=x.device)
    for c0 in range(0, cin, _BLOCK):
        xb = xr if cin <= _BLOCK else xr[:, c0:c0 + _BLOCK].contiguous()
        for o0 in range(0, cout, _BLOCK):
            db = dyr if cout <= _BLOCK else dyr[:, o0:o0 + _BLOCK].contiguous()
            dw[:, c0:c0 + xb.shape[1], o0:o0 + db.shape[1]] = sparse._conv_wgrad_raw(xb, db, pairs, K, 'dense_strided_wgrad')
    if kind == 'conv':
        return dw.permute(2, 1, 0).reshape(cout, cin, kh, kw)
    return dw.permute(1, 2, 0).reshape(cin, cout, kh, kw)


def supported(conv, x):
    import torch.nn as nn
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        return False
    if conv.bias is not None or conv.groups != 1 or conv.dilation != (1, 1) or conv.kernel_size[0] * conv.kernel_size[1] > 32:
        return False
    if conv.stride[0] != conv.stride[1] or getattr(conv, 'padding_mode', 'zeros') != 'zeros':
        return False
    if isinstance(conv, nn.ConvTranspose2d):
        if conv.padding != (0, 0) or conv.output_padding != (0, 0):
            return False
    cin, cout = conv.in_channels, conv.out_channels
    ok = lambda c: c % _BLOCK == 0 or c in (16, 32, 64)
    blk = lambda c: min(c, _BLOCK)
    return ok(cin) and ok(cout) and bool(sparse.lib.crb_sparse_conv_supported(blk(cin), blk(cout)))


class _vendor_default_solvers(object):
    """the vendor forward / data-gradient kernels of these two layers give the same bits run to run in the default mode (only the weight
    gradient did not: tools/dbg_determinism.py). Under the deterministic flag MIOpen falls back to its naive reference solver for them:
    157 ms per call at 16 x 200 x 176 instead of 0.6 - so the flag is lowered around exactly these calls."""

    def __enter__(self):
        self.was = torch.are_deterministic_algorithms_enabled()
        self.warn = torch.is_deterministic_algorithms_warn_only_enabled()
        self.cudnn = torch.backends.cudnn.deterministic
        torch.use_deterministic_algorithms(False)
        torch.backends.cudnn.deterministic = False

    def __exit__(self, *exc):
        torch.backends.cudnn.deterministic = self.cudnn
        torch.use_deterministic_algorithms(self.was, warn_only=self.warn)
        return False


class _StridedConvDet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, kind, s, p):
        ctx.save_for_backward(x, w)
        ctx.kind, ctx.s, ctx.p = kind, s, p
        with _vendor_default_solvers():
            if kind == 'conv':
                return torch.nn.functional.conv2d(x, w, None, s, p)
            return torch.nn.functional.conv_transpose2d(x, w, None, s, 0)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        kind, s, p = ctx.kind, ctx.s, ctx.p
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            with _vendor_default_solvers():
                dx = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p] if kind == 'conv' else [0, 0], [1, 1],
                                                         kind == 'deconv', [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            dw = weight_grad(kind, x, dy, w.shape[2], w.shape[3], s, p)
        return dx, dw, None, None, None


def conv_det(conv, x, padding=None):
    """conv(x) with the deterministic weight gradient (padding overrides conv.padding: the folded ZeroPad2d)"""
    import torch.nn as nn
    if isinstance(conv, nn.ConvTranspose2d):
        return _StridedConvDet.apply(x, conv.weight, 'deconv', conv.stride[0], 0)
    pd = conv.padding if padding is None else padding
    if pd[0] != pd[1]:
        raise ValueError('conv_det: square padding only')
    return _StridedConvDet.apply(x, conv.weight, 'conv', conv.stride[0], int(pd[0]))
