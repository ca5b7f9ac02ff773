"""3x3 stride-1 pad-1 convolution on channels_last maps as Winograd F(2x2,3x3) on the f32 MFMA: 2.25x fewer multiplications
than the direct convolution, results equal to it up to f32 rounding of the transforms (4e-7 of the output scale against an f64
convolution; MIOpen's implicit GEMM: 1.2e-6).

Kernels: forward and input gradient run on the round-6 split-bf16 kernel (crb_conv3x3_winograd4_nhwc; KERNEL below) where it has
an instance, otherwise on the round-4 f32-MFMA design (crb_conv3x3_winograd2_nhwc; 0.71 ms per 128->128 @ 16x200x176 call against
MIOpen's 1.41), which also is what the `*2` names meant before round 6; the BEV backbone calls — `conv3x3` (training: forward + input gradient) and `fold` / `conv3x3_U2`
(inference: BatchNorm folded, bias + ReLU in the epilogue); the round-3 kernel (crb_conv3x3_winograd_nhwc, 1.15 ms) stays for
A/B runs in tools/."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda, CrbHipError


# When set to a list, every Winograd launch appends (kind, cin, cout, N, H, W, ev0, ev1) with HIP events on the launch stream
# (torch's current stream IS the stream handed to the C-ABI); bench.py reads them for the roofline / kernel table.
# kind: 'wino_conv' (forward and input gradient: the same kernel) | 'wino_wgrad' (both launches of crb_winograd2_wgrad)
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(e0, *rec):
    if e0 is not None and PROFILE is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append(rec + (e0, e1))


def supported(cin, cout):
    """first design (round 3): measurement library only since round 5"""
    from ._lib import MEASURE
    return bool(MEASURE and lib.crb_winograd_supported(int(cin), int(cout)))


def transform_weights(g):
    """g (3,3,Cin,Cout) contiguous [ky][kx][ci][co] -> U (16,Cin,Cout)"""
    from ._lib import require_measure
    require_measure('crb_winograd_weights')
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


# Which kernel runs the forward / input-gradient convolutions: 'x6' = csrc/winograd_conv4.hip (the 16 GEMMs as six bf16 MFMA passes over an
# exact three-way split of the f32 operands: f32 in, f32 out, errors against f64 at the f32-MFMA kernel's level, ~11 % faster), 'f32' =
# csrc/winograd_conv2.hip (exact-f32 MFMA). 'x6' needs Cin % 16 == 0, Cout % 64 == 0 and maps of at least 31 rows; everything else
# (and everything under CRB_WINOGRAD_KERNEL=f32) runs on the f32-MFMA kernel. The weight gradient is the f32-MFMA kernel's either way.
KERNEL = __import__('os').environ.get('CRB_WINOGRAD_KERNEL', 'x6')


def _use4(kin, kout):
    return KERNEL == 'x6' and bool(lib.crb_winograd4_supported(int(kin), int(kout), 31, 1))


# the split-bf16 kernel's third form (workgroup tile 32 tiles x 128 channels: crb_conv3x3_winograd4c_nhwc) where it has an instance
# (Cout % 128 == 0); CRB_WINOGRAD_FORM_C=0: the 64 x 64 form everywhere (A/B). Same outputs, bit for bit; another image layout.
FORM_C = __import__('os').environ.get('CRB_WINOGRAD_FORM_C', '1') == '1'


def _use_c(kin, kout):
    return FORM_C and _use4(kin, kout) and bool(lib.crb_winograd4c_supported(int(kin), int(kout), 31, 1))


# images made ahead by prepare_weights2 (one launch for all layers of a step): key -> image. A key names the weight's memory, its
# autograd version (an optimizer step bumps it) and the mode: a stale image cannot be returned
_PREPARED = {}
PREPARE = __import__('os').environ.get('CRB_WINOGRAD_PREPARE', '1') == '1'


def _prep_key(w, mode):
    return (w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride()), w.device.index, KERNEL, FORM_C, int(mode))


def prepare_weights2(weights, input_grad=True):
    """the forward (and input-gradient) images of all `weights` (nn.Conv2d weights (Cout,Cin,3,3) f32 on one device) in ONE launch
    (crb_winograd2_weights_conv_multi) instead of one ~10 us launch per image; weights_forward2 / weights_input_grad2 of the same
    tensors (same memory, same version) then return the prepared images"""
    import ctypes
    _PREPARED.clear()
    jobs = []
    for w in weights:
        if not (PREPARE and w.is_cuda and w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (3, 3)):
            continue
        for mode in ((0, 1) if input_grad else (0,)):
            cout, cin = w.shape[0], w.shape[1]
            if lib.crb_winograd2_supported(int(cout if mode else cin), int(cin if mode else cout), 5, 1):
                jobs.append((w.detach(), mode))
    def kshape(w, mode):
        return (w.shape[0], w.shape[1]) if mode else (w.shape[1], w.shape[0])
    jobs4 = [(w, m) for w, m in jobs if _use4(*kshape(w, m))]
    jobs2 = [(w, m) for w, m in jobs if not _use4(*kshape(w, m))]
    for four, todo in ((True, jobs4), (False, jobs2)):
        for lo in range(0, len(todo), 32):
            part = todo[lo:lo + 32]
            n = len(part)
            if four:
                Us = [torch.empty((int(lib.crb_winograd4_weights_bytes(w.shape[1], w.shape[0])),), dtype=torch.uint8, device=w.device)
                      for w, _ in part]
            else:
                Us = [torch.empty((16 * w.shape[0] * w.shape[1],), dtype=torch.float32, device=w.device) for w, _ in part]
            wp = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _ in part])
            up = (ctypes.c_void_p * n)(*[u.data_ptr() for u in Us])
            st = (ctypes.c_int64 * (4 * n))(*[v for w, _ in part for v in w.stride()])
            ci = (ctypes.c_int32 * n)(*[w.shape[1] for w, _ in part])
            co = (ctypes.c_int32 * n)(*[w.shape[0] for w, _ in part])
            md = (ctypes.c_int32 * n)(*[(m + 2 if (four and _use_c(*kshape(w, m))) else m) for w, m in part])
            fn = lib.crb_winograd4_weights_conv_multi if four else lib.crb_winograd2_weights_conv_multi
            check(fn(n, wp, st, up, ci, co, md, cur_stream(part[0][0].device)),
                  'crb_winograd4_weights_conv_multi' if four else 'crb_winograd2_weights_conv_multi')
            for (w, mode), U in zip(part, Us):
                if four:
                    U.wino4_shape = kshape(w, mode)
                    U._crb_mode = mode
                    U._crb_c = _use_c(*kshape(w, mode))
                else:
                    U.wino2_shape = kshape(w, mode)
                U._crb_src = w              # (keeps the weight's storage alive: no other tensor can take the key's address meanwhile)
                _PREPARED[_prep_key(w, mode)] = U
    return len(jobs)


def forget_prepared_forward():
    """the forward images were for the pass that prepared them; the input-gradient images stay for its backward (autograd refuses a
    backward whose saved weight changed version)"""
    for k in [k for k in _PREPARED if k[-1] == 0]:
        del _PREPARED[k]


def _weights_conv2(weight, mode):
    """nn.Conv2d weight (Cout,Cin,3,3), any strides -> image of the forward (mode 0) / input-gradient (mode 1) convolution"""
    require_cuda(weight)
    if _PREPARED:
        hit = _PREPARED.get(_prep_key(weight, mode))
        if hit is not None:
            return hit
    cout, cin = weight.shape[0], weight.shape[1]
    if _use4(cout if mode else cin, cin if mode else cout):
        return _weights_conv4(weight, mode)
    return _weights_conv2_f32(weight, mode)


def _weights_conv2_f32(weight, mode):
    """the f32-MFMA kernel's image (crb_winograd2_weights_conv)"""
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


def _f32_image_of(U):
    """the f32-MFMA kernel's image of the weight a split-bf16 image was made from (maps with fewer than 31 rows), kept on the image"""
    U2 = getattr(U, '_crb_f32', None)
    if U2 is None:
        U2 = U._crb_f32 = _weights_conv2_f32(U._crb_src, U._crb_mode)
    return U2


def weights_forward2(weight):
    return _weights_conv2(weight, 0)


def weights_input_grad2(weight):
    return _weights_conv2(weight, 1)


def conv3x3_U2(x, U, bias=None, relu=False):
    """x (N,Cin,H,W) f32 channels_last, U = weights_forward2(...) -> y (N,Cout,H,W) channels_last (second kernel)"""
    require_cuda(x, U)
    if hasattr(U, 'wino4_shape'):
        if supported4(U.wino4_shape[0], U.wino4_shape[1], x.shape[2], x.shape[3]):
            return conv3x3_U4(x, U, bias, relu)
        U = _f32_image_of(U)
    xv = _nhwc(x.float())
    N, H, W, cin = xv.shape
    ucin, cout = U.wino2_shape
    if ucin != cin or not supported2(cin, cout, H, W):
        raise CrbHipError('no Winograd (2) instance for %d -> %d channels on a %d x %d map' % (cin, cout, H, W))
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    e0 = _prof_begin()
    check(lib.crb_conv3x3_winograd2_nhwc(xv.data_ptr(), ptr(U), y.data_ptr(), N, H, W, cin, cout,
                                         ptr(bias.contiguous().float()) if bias is not None else None, int(bool(relu)),
                                         cur_stream(x.device)), 'crb_conv3x3_winograd2_nhwc')
    _prof_end(e0, 'wino_conv', cin, cout, N, H, W)
    return y


# ---- round 6: the same convolution on the bf16 matrix pipe through an exact three-way split of the f32 operands
#      (csrc/winograd_conv4.hip, crb_conv3x3_winograd4_nhwc): f32 in, f32 out, errors against f64 at the level of the f32-MFMA kernel

def supported4(cin, cout, H, W):
    return bool(lib.crb_winograd4_supported(int(cin), int(cout), int(H), int(W)))


def _weights_conv4(weight, mode):
    """nn.Conv2d weight (Cout,Cin,3,3), any strides -> split-bf16 image of the forward (mode 0) / input-gradient (mode 1) convolution"""
    require_cuda(weight)
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    cout, cin = w.shape[0], w.shape[1]
    U = torch.empty((int(lib.crb_winograd4_weights_bytes(cin, cout)),), dtype=torch.uint8, device=w.device)
    so, si, sky, skx = w.stride()
    form_c = _use_c(cout if mode else cin, cin if mode else cout)
    check(lib.crb_winograd4_weights_conv(w.data_ptr(), so, si, sky, skx, ptr(U), cin, cout, mode + (2 if form_c else 0),
                                         cur_stream(w.device)), 'crb_winograd4_weights_conv')
    U.wino4_shape = (cout, cin) if mode else (cin, cout)
    U._crb_src, U._crb_mode, U._crb_c = w, mode, form_c
    return U


def weights_forward4(weight):
    return _weights_conv4(weight, 0)


def weights_input_grad4(weight):
    return _weights_conv4(weight, 1)


def conv3x3_U4(x, U, bias=None, relu=False):
    """x (N,Cin,H,W) f32 channels_last, U = weights_forward4(...) -> y (N,Cout,H,W) channels_last"""
    require_cuda(x, U)
    xv = _nhwc(x.float())
    N, H, W, cin = xv.shape
    ucin, cout = U.wino4_shape
    if ucin != cin or not supported4(cin, cout, H, W):
        raise CrbHipError('no Winograd (4) instance for %d -> %d channels on a %d x %d map' % (cin, cout, H, W))
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    e0 = _prof_begin()
    fn = lib.crb_conv3x3_winograd4c_nhwc if getattr(U, '_crb_c', False) else lib.crb_conv3x3_winograd4_nhwc
    check(fn(xv.data_ptr(), ptr(U), y.data_ptr(), N, H, W, cin, cout,
             ptr(bias.contiguous().float()) if bias is not None else None, int(bool(relu)),
             cur_stream(x.device)), 'crb_conv3x3_winograd4_nhwc')
    _prof_end(e0, 'wino_conv', cin, cout, N, H, W)
    return y


def conv3x3_stats_U4(x, U):
    """(y, slab sums of y and y^2) of the bias-free convolution (crb_conv3x3_winograd4_stats_nhwc)"""
    require_cuda(x, U)
    xv = _nhwc(x.float())
    N, H, W, cin = xv.shape
    ucin, cout = U.wino4_shape
    if ucin != cin or not supported4(cin, cout, H, W):
        raise CrbHipError('no Winograd (4) instance for %d -> %d channels on a %d x %d map' % (cin, cout, H, W))
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    form_c = getattr(U, '_crb_c', False)
    slabs = lib.crb_winograd4c_stats_slabs(N, H, W) if form_c else lib.crb_winograd4_stats_slabs(N, H, W)
    stats = torch.empty((int(slabs), 2, cout), dtype=torch.float32, device=x.device)
    e0 = _prof_begin()
    fn = lib.crb_conv3x3_winograd4c_stats_nhwc if form_c else lib.crb_conv3x3_winograd4_stats_nhwc
    check(fn(xv.data_ptr(), ptr(U), y.data_ptr(), ptr(stats), N, H, W, cin, cout, cur_stream(x.device)),
          'crb_conv3x3_winograd4_stats_nhwc')
    _prof_end(e0, 'wino_conv', cin, cout, N, H, W)
    return y, stats


def _nhwc(x):
    """(N,C,H,W) tensor in channels_last memory -> its (N,H,W,C) view (no copy); other layouts are converted"""
    if not x.is_contiguous(memory_format=torch.channels_last):
        x = x.contiguous(memory_format=torch.channels_last)
    return x.permute(0, 2, 3, 1)


def conv3x3_U(x, U, bias=None, relu=False):
    """x (N,Cin,H,W) f32 channels_last, U (16,Cin,Cout) -> y (N,Cout,H,W) channels_last (first design: measurement library)"""
    from ._lib import require_measure
    require_measure('crb_conv3x3_winograd_nhwc')
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


def wgrad_supported(cin, cout, H, W):
    return bool(lib.crb_winograd2_wgrad_supported(int(cin), int(cout), int(H), int(W)))


_WGRAD_WS = {}


def conv3x3_wgrad(x, dy, like):
    """x (N,Cin,H,W), dy (N,Cout,H,W) f32 channels_last -> gradient of the nn.Conv2d weight (Cout,Cin,3,3) in the memory layout
    of `like` (crb_winograd2_wgrad: both operands transformed inside the kernel, dW = G^T dU G)"""
    require_cuda(x, dy)
    xv, gv = _nhwc(x.float()), _nhwc(dy.float())
    N, H, W, cin = xv.shape
    cout = gv.shape[3]
    if gv.shape[:3] != xv.shape[:3] or not wgrad_supported(cin, cout, H, W):
        raise CrbHipError('no Winograd weight-gradient instance for %d -> %d channels' % (cin, cout))
    dw = torch.empty_like(like, dtype=torch.float32)
    nbytes = int(lib.crb_winograd2_wgrad_workspace_bytes(cin, cout))
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    ws = _WGRAD_WS.get(key)                      # one per (device, stream): every call on a stream is ordered behind the last
    if ws is None or ws.numel() * 4 < nbytes:
        ws = _WGRAD_WS[key] = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
    so, si, sky, skx = dw.stride()
    e0 = _prof_begin()
    check(lib.crb_winograd2_wgrad(xv.data_ptr(), gv.data_ptr(), dw.data_ptr(), so, si, sky, skx, N, H, W, cin, cout, ptr(ws),
                                  ws.numel() * 4, cur_stream(x.device)), 'crb_winograd2_wgrad')
    _prof_end(e0, 'wino_wgrad', cin, cout, N, H, W)
    return dw


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
        wino_dw = WGRAD and ctx.needs_input_grad[1] and wgrad_supported(weight.shape[1], weight.shape[0], x.shape[2], x.shape[3])
        if wino_dw:
            dw = conv3x3_wgrad(x, dy, weight)
        if (ctx.needs_input_grad[1] and not wino_dw) or (ctx.needs_input_grad[0] and not wino_dx):
            gi, gw, _ = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                            [ctx.needs_input_grad[0] and not wino_dx,
                                                             ctx.needs_input_grad[1] and not wino_dw, False])
            dw = gw if (ctx.needs_input_grad[1] and not wino_dw) else dw
            dx = gi if (ctx.needs_input_grad[0] and not wino_dx) else dx
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db


WGRAD = __import__('os').environ.get('CRB_WINOGRAD_WGRAD', '1') != '0'      # 0: weight gradients on MIOpen (A/B)


def conv3x3(x, weight, bias=None):
    """differentiable 3x3 stride-1 pad-1 convolution: forward, input gradient and weight gradient on the Winograd kernels
    (weight gradient on MIOpen where crb_winograd2_wgrad has no instance: channel counts not multiples of 64).
    Callers check `supported2(Cin, Cout, H, W)` first."""
    return _Conv3x3.apply(x, weight, bias)


# statistics of the following BatchNorm from the convolution's epilogue (training): CRB_WINOGRAD_STATS=0 switches it off (A/B)
STATS = __import__('os').environ.get('CRB_WINOGRAD_STATS', '1') != '0'


class _Conv3x3Stats(torch.autograd.Function):
    """_Conv3x3 without bias whose forward also returns the slab sums of y and y^2 (crb_conv3x3_winograd2_stats_nhwc) for
    crb_bn_relu_forward_partials: the BatchNorm that follows launches no statistics pass over y"""

    @staticmethod
    def forward(ctx, x, weight):
        require_cuda(x, weight)
        ctx.save_for_backward(x, weight)
        xv = _nhwc(x.float())
        N, H, W, cin = xv.shape
        cout = weight.shape[0]
        U = weights_forward2(weight)
        if hasattr(U, 'wino4_shape') and supported4(cin, cout, H, W):
            y, stats = conv3x3_stats_U4(x, U)
        else:
            if hasattr(U, 'wino4_shape'):
                U = _f32_image_of(U)
            y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            stats = torch.empty((int(lib.crb_winograd2_stats_slabs(N, H, W)), 2, cout), dtype=torch.float32, device=x.device)
            e0 = _prof_begin()
            check(lib.crb_conv3x3_winograd2_stats_nhwc(xv.data_ptr(), ptr(U), y.data_ptr(), ptr(stats), N, H, W, cin, cout,
                                                       cur_stream(x.device)), 'crb_conv3x3_winograd2_stats_nhwc')
            _prof_end(e0, 'wino_conv', cin, cout, N, H, W)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)          # (no zero tensor for the statistics output's gradient in every backward call)
        return y, stats

    @staticmethod
    def backward(ctx, dy, _ds):
        x, weight = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if supported2(weight.shape[0], weight.shape[1], x.shape[2], x.shape[3]):
                dx = conv3x3_U2(dy, weights_input_grad2(weight))
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            if WGRAD and wgrad_supported(weight.shape[1], weight.shape[0], x.shape[2], x.shape[3]):
                dw = conv3x3_wgrad(x, dy, weight)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return dx, dw


def conv3x3_stats(x, weight):
    """(y, slab sums) of the bias-free 3x3 stride-1 pad-1 convolution; pass the sums to crbhip.bnrelu.bn_relu(..., slabs=)"""
    return _Conv3x3Stats.apply(x, weight)
