"""3x3 stride-1 pad-1 convolution on channels_last maps as Winograd F(2x2,3x3) on the f32 MFMA: 2.25x fewer multiplications
than the direct convolution, results equal to it up to f32 rounding of the transforms (4e-7 of the output scale against an f64
convolution; MIOpen's implicit GEMM: 1.2e-6).

Two kernels: the round-4 design (crb_conv3x3_winograd2_nhwc, `*2` functions; 0.77 ms per 128->128 @ 16x200x176 call against
MIOpen's 1.41) is what the BEV backbone runs by default — `conv3x3` (training: forward + input gradient) and `fold` / `conv3x3_U2`
(inference: BatchNorm folded, bias + ReLU in the epilogue); the round-3 kernel (crb_conv3x3_winograd_nhwc, 1.15 ms) stays for
A/B runs in tools/."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, CrbHipError


def supported(cin, cout):
    return bool(lib.crb_winograd_supported(int(cin), int(cout)))


def transform_weights(g):
    """g (3,3,Cin,Cout) contiguous [ky][kx][ci][co] -> U (16,Cin,Cout)"""
    require_cuda(g)
    cin, cout = g.shape[2], g.shape[3]
    U = torch.empty((16, cin, cout), dtype=torch.float32, device=g.device)
    check(lib.crb_winograd_weights(ptr(g.contiguous().float()), ptr(U), cin, cout, cur_stream(g.device)), 'crb_winograd_weights')
    return U


def weights_forward(weight):
    """nn.Conv2d weight (Cout,Cin,3,3) -> U of the forward convolution"""
    return transform_weights(weight.permute(2, 3, 1, 0).contiguous())


def weights_input_grad(weight):
    """nn.Conv2d weight (Cout,Cin,3,3) -> U of the convolution that maps dy (Cout channels) to dx (Cin channels):
    g'[ky][kx][co][ci] = w[co][ci][2-ky][2-kx]"""
    return transform_weights(weight.flip(2, 3).permute(2, 3, 0, 1).contiguous())


def supported2(cin, cout, H, W):
    return bool(lib.crb_winograd2_supported(int(cin), int(cout), int(H), int(W)))


def transform_weights2(g):
    """g (3,3,Cin,Cout) contiguous [ky][kx][ci][co] -> weight image of the second kernel (crb_winograd2_weights), tagged with
    its channel counts"""
    require_cuda(g)
    cin, cout = g.shape[2], g.shape[3]
    U = torch.empty((16 * cin * cout,), dtype=torch.float32, device=g.device)
    check(lib.crb_winograd2_weights(ptr(g.contiguous().float()), ptr(U), cin, cout, cur_stream(g.device)), 'crb_winograd2_weights')
    U.wino2_shape = (cin, cout)
    return U


def _weights_conv2(weight, mode):
    """nn.Conv2d weight (Cout,Cin,3,3), any strides -> image of the forward (mode 0) / input-gradient (mode 1) convolution"""
    require_cuda(weight)
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    cout, cin = w.shape[0], w.shape[1]
    U = torch.empty((16 * cin * cout,), dtype=torch.float32, device=w.device)
    so, si, sky, skx = w.stride()
    check(lib.crb_winograd2_weights_conv(w.data_ptr(), so, si, sky, skx, ptr(U), cin, cout, mode, cur_stream(w.device)),   # strided: no ptr()
          'crb_winograd2_weights_conv')
    U.wino2_shape = (cout, cin) if mode else (cin, cout)
    return U


def weights_forward2(weight):
    return _weights_conv2(weight, 0)


def weights_input_grad2(weight):
    return _weights_conv2(weight, 1)


def conv3x3_U2(x, U, bias=None, relu=False):
    """x (N,Cin,H,W) f32 channels_last, U = weights_forward2(...) -> y (N,Cout,H,W) channels_last (second kernel)"""
    require_cuda(x, U)
    xv = _nhwc(x.float())
    N, H, W, cin = xv.shape
    ucin, cout = U.wino2_shape
    if ucin != cin or not supported2(cin, cout, H, W):
        raise CrbHipError('no Winograd (2) instance for %d -> %d channels on a %d x %d map' % (cin, cout, H, W))
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    check(lib.crb_conv3x3_winograd2_nhwc(xv.data_ptr(), ptr(U), y.data_ptr(), N, H, W, cin, cout,
                                         ptr(bias.contiguous().float()) if bias is not None else None, int(bool(relu)),
                                         cur_stream(x.device)), 'crb_conv3x3_winograd2_nhwc')
    return y


def _nhwc(x):
    """(N,C,H,W) tensor in channels_last memory -> its (N,H,W,C) view (no copy); other layouts are converted"""
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x.permute(0, 2, 3, 1)


def conv3x3_U(x, U, bias=None, relu=False):
    """x (N,Cin,H,W) f32 channels_last, U (16,Cin,Cout) -> y (N,Cout,H,W) channels_last"""
    require_cuda(x, U)
    xv = _nhwc(x.float())
    N, H, W, cin = xv.shape
    cout = U.shape[2]
    if U.shape[1] != cin or not supported(cin, cout):
        raise CrbHipError('no Winograd instance for %d -> %d channels' % (cin, cout))
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    check(lib.crb_conv3x3_winograd_nhwc(xv.data_ptr(), ptr(U), y.data_ptr(), N, H, W, cin, cout,
                                        ptr(bias.contiguous().float()) if bias is not None else None, int(bool(relu)),
                                        cur_stream(x.device)), 'crb_conv3x3_winograd_nhwc')
    return y


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return conv3x3_U2(x, weights_forward2(weight), bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        wino_dx = ctx.needs_input_grad[0] and supported2(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3])
        if wino_dx:
            dx = conv3x3_U2(dy, weights_input_grad2(weight))
        if ctx.needs_input_grad[1] or (ctx.needs_input_grad[0] and not wino_dx):
            gi, gw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [ctx.needs_input_grad[0] and not wino_dx,
                                                             ctx.needs_input_grad[1], False])
            dw = gw if ctx.needs_input_grad[1] else None
            dx = gi if (ctx.needs_input_grad[0] and not wino_dx) else dx
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db


def conv3x3(x, weight, bias=None):
    """differentiable 3x3 stride-1 pad-1 convolution: forward and input gradient on the Winograd kernel, weight gradient on
    MIOpen (aten.convolution_backward). Callers check `supported2(Cin, Cout, H, W)` first."""
    return _Conv3x3.apply(x, weight, bias)
