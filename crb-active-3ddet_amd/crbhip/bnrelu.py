"""Host side of the fused sparse BatchNorm1d(+ReLU) kernels (C-ABI: crb_bn_relu_*)."""
import torch

from ._lib import lib, check, ptr, cur_stream, require_cuda


FUSE_RUNNING = __import__('os').environ.get('CRB_BN_FUSE_RUNNING', '1') == '1'
# finalize inside the statistics launch ("last block done" tickets, include/crb_hip.h): 2 launches per call instead of 3
TICKETS = __import__('os').environ.get('CRB_BN_TICKETS', '1') == '1'
_ticket_areas = {}


def _scratch(dev, wsb):
    """(workspace, ticket area or None) for one crb_bn_* call on the current stream of `dev`. The ticket area is one zeroed
    int32 tensor per (device, stream), handed to every call enqueued on that stream: the kernels leave it zero and stream
    order keeps consecutive calls apart."""
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    if not TICKETS:
        return ws, None
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    t = _ticket_areas.get(key)
    if t is None:
        t = _ticket_areas[key] = torch.zeros((lib.crb_bn_ticket_ints(),), dtype=torch.int32, device=dev)
    return ws, t


def _bn_check(rc, what):
    """check() for the calls that were handed a ticket area: the kernels leave the area zero only when they ran to the end. After
    a failed / skipped launch the counters may be anything and every later call on the stream would finalize too early or never
    (ADVICE r03): drop the areas, the next call starts from fresh zeroed ones."""
    if rc != 0:
        _ticket_areas.clear()
    check(rc, what)


# ---- per-frame statistics ------------------------------------------------------------------------------------------
# CRB stage 2 runs the detector in train mode on ONE frame at a time (crb_sampling.py:174-212): every BatchNorm layer
# normalises with that frame's own statistics. To batch G frames per pass with the same values, the training-mode entry
# points below split their rows into the frames' row ranges and run the SAME kernels once per range while a `frame_groups`
# context is active (ranges: G equal parts, or the offsets noted for a tensor whose rows are ragged per frame — sparse
# voxel features). Running statistics are updated once per range, in frame order, exactly like G separate passes.
_GROUPS = None


class frame_groups(object):
    def __init__(self, G):
        self.G = int(G)
        self.ragged = {}

    def __enter__(self):
        global _GROUPS
        self.prev, _GROUPS = _GROUPS, self
        return self

    def __exit__(self, *exc):
        global _GROUPS
        _GROUPS = self.prev
        return False

    def note_rows(self, t, offsets):
        """rows of tensor t belong to the frames by these G+1 host offsets (not G equal parts). Keyed on the tensor OBJECT
        (weak reference): the caching allocator hands a freed tensor's address to later ones, an address key would let an
        unrelated tensor with the same row count inherit stale frame boundaries"""
        if offsets is not None:
            import weakref
            assert len(offsets) == self.G + 1 and offsets[-1] == t.shape[0], (len(offsets), offsets[-1], t.shape)
            if len(self.ragged) > 64:                       # drop entries whose tensor is gone
                self.ragged = {k: v for k, v in self.ragged.items() if v[0]() is not None}
            self.ragged[id(t)] = (weakref.ref(t), [int(v) for v in offsets])

    def noted(self, t):
        """the ragged offsets noted for exactly this tensor object, or None"""
        e = self.ragged.get(id(t))
        return e[1] if (e is not None and e[0]() is t and e[1][-1] == t.shape[0]) else None

    def offsets(self, t, n_units=None):
        """row ranges of t per frame; n_units: t has this many equal units per row block (rows = units * k)"""
        r = self.noted(t)
        if r is not None:
            return r
        n = t.shape[0]
        if n % self.G:
            raise ValueError('%d rows do not split into %d frames: note_rows() the ragged offsets first' % (n, self.G))
        return [g * (n // self.G) for g in range(self.G + 1)]


def active_groups():
    return _GROUPS


def _frames_forward(x, off, bn, relu, out=None, col=0):
    """no-grad training BatchNorm(+ReLU) of x (n,C) with per-frame statistics (row ranges `off`), one C-ABI call; written
    into columns [col, col+C) of `out` (n, W) when given"""
    import ctypes
    x = x.contiguous()
    n, C = x.shape
    z = torch.empty_like(x) if out is None else out
    ld = 0 if out is None else out.shape[1]
    arr = (ctypes.c_int64 * len(off))(*[int(v) for v in off])
    wsb = lib.crb_bn_frames_workspace_bytes(len(off) - 1, max(b - a for a, b in zip(off[:-1], off[1:])), C)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
    with torch.no_grad():
        bn.num_batches_tracked += len(off) - 1
    zp = ctypes.c_void_p(z.data_ptr() + 4 * col)
    check(lib.crb_bn_relu_forward_frames(ptr(x), len(off) - 1, arr, C, ptr(bn.weight.contiguous().float()),
                                         ptr(bn.bias.contiguous().float()), float(bn.eps), int(relu), zp, ld,
                                         ptr(bn.running_mean), ptr(bn.running_var), float(bn.momentum), ptr(ws), wsb,
                                         cur_stream(x.device)), 'crb_bn_relu_forward_frames')
    _touch(bn.running_mean, bn.running_var)
    return z


def frame_groups_active():
    """inside a frame_groups(G > 1) context: BatchNorm layers keep per-frame statistics (batched CRB stage 2)"""
    return _GROUPS is not None and _GROUPS.G > 1


def supported(x, bn):
    C = x.shape[1]
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2 and C % 4 == 0 and
            256 % (C // 4) == 0 and bn.affine and bn.track_running_stats)


# When set to a list, the training BatchNorm(+ReLU) op appends (kind, n rows, C, ev0, ev1) with HIP events on the launch stream
# (kind 'bn_fwd' | 'bn_bwd'); bench.py reads them for the kernel table (algorithmic bytes: 3 / 5 passes of 4 n C).
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


class _BNReLUTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, running_mean=None, running_var=None, momentum=0.0, nbt=None, slabs=None):
        require_cuda(x, gamma, beta)
        ctx.set_materialize_grads(False)          # mean / var carry no gradient: no zero tensors made for them in backward
        x = x.contiguous()
        n, C = x.shape
        dev = x.device
        z = torch.empty_like(x)
        mean = torch.empty((C,), dtype=torch.float32, device=dev)
        var = torch.empty_like(mean)
        invstd = torch.empty_like(mean)
        wsb = lib.crb_bn_workspace_bytes(n, C)
        ws, tk = _scratch(dev, wsb)
        g, b = gamma.contiguous().float(), beta.contiguous().float()
        e0 = _prof_begin()
        if slabs is not None:
            # statistics from the slab sums the producer of x wrote (the Winograd forward kernel's epilogue): no pass over x for them
            sl = slabs.contiguous()
            _bn_check(lib.crb_bn_relu_forward_partials(ptr(x), n, C, ptr(sl), sl.shape[0], ptr(g), ptr(b), float(eps), int(relu),
                                                       ptr(z), 0, ptr(mean), ptr(var), ptr(invstd), ptr(running_mean),
                                                       ptr(running_var), ptr(nbt), float(momentum), ptr(ws), wsb, ptr(tk),
                                                       cur_stream(dev)), 'crb_bn_relu_forward_partials')
            _prof_end(e0, 'bn_apply', n, C)
        else:
            _bn_check(lib.crb_bn_relu_forward(ptr(x), n, C, ptr(g), ptr(b), float(eps), int(relu), ptr(z), 0, ptr(mean), ptr(var),
                                              ptr(invstd), ptr(running_mean), ptr(running_var), ptr(nbt), float(momentum), ptr(ws),
                                              wsb, ptr(tk), cur_stream(dev)), 'crb_bn_relu_forward')
            _prof_end(e0, 'bn_fwd', n, C)
        _touch(running_mean, running_var, nbt)
        ctx.save_for_backward(x, mean, invstd, g, b)
        ctx.relu = int(relu)
        ctx.mark_non_differentiable(mean, var)
        return z, mean, var

    @staticmethod
    def backward(ctx, dz, _dm, _dv):
        x, mean, invstd, g, b = ctx.saved_tensors
        n, C = x.shape
        dev = x.device
        dz = dz.contiguous().float()
        dx = torch.empty_like(x)
        dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        wsb = lib.crb_bn_workspace_bytes(n, C)
        ws, tk = _scratch(dev, wsb)
        e0 = _prof_begin()
        _bn_check(lib.crb_bn_relu_backward(ptr(x), ptr(dz), 0, n, C, ptr(mean), ptr(invstd), ptr(g), ptr(b), ctx.relu, ptr(dx),
                                       ptr(dgamma), ptr(dbeta), ptr(ws), wsb, ptr(tk), cur_stream(dev)), 'crb_bn_relu_backward')
        _prof_end(e0, 'bn_bwd', n, C)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def _touch(*tensors):
    """tensors a kernel wrote through their raw pointers (running statistics, the batch counter): bump their version counters
    as an in-place torch op would — the eval-time caches (folded Conv+BN weights, rsqrt(running_var + eps)) are keyed on them"""
    ts = [t for t in tensors if t is not None]
    if not ts:
        return
    try:
        torch._C._autograd._unsafe_set_version_counter(ts, [t._version + 1 for t in ts])
    except (TypeError, AttributeError):             # other torch builds: (tensor, int) signature or no such private hook
        for t in ts:
            try:
                torch._C._autograd._unsafe_set_version_counter(t, t._version + 1)
            except (TypeError, AttributeError):
                with torch.no_grad():
                    t.add_(0)                       # one tiny launch per buffer, same effect on the version counter


def _invstd(bn):
    """rsqrt(running_var + eps) of an eval-mode BatchNorm, cached on the module and keyed on the buffer's address and version
    (load_state_dict / a training step / .to() change the key): two tiny launches per layer and batch otherwise"""
    rv = bn.running_var
    key = (rv.data_ptr(), rv._version, float(bn.eps))
    hit = bn.__dict__.get('_crb_invstd')
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        val = torch.rsqrt(rv + bn.eps)
    bn.__dict__['_crb_invstd'] = (key, val)
    return val


def _counter(bn):
    """bn.num_batches_tracked for the kernels to increment, or None after incrementing it here (not an int64 device scalar)"""
    t = bn.num_batches_tracked
    if t is not None and t.is_cuda and t.dtype == torch.int64 and t.numel() == 1:
        return t
    if t is not None:
        with torch.no_grad():
            t += 1
    return None


def bn_relu(x, bn, relu=True, slabs=None):
    """x (N,C) cuda f32; bn: nn.BatchNorm1d. Same semantics as relu(bn(x)) incl. the running-statistics update
    (momentum, unbiased running variance, num_batches_tracked)."""
    n, C = x.shape
    if bn.training and _GROUPS is not None and _GROUPS.G > 1:
        off, grp = _GROUPS.offsets(x), _GROUPS
        if not torch.is_grad_enabled() and bn.momentum is not None and min(b - a for a, b in zip(off[:-1], off[1:])) >= 2:
            return _frames_forward(x, off, bn, relu)
        with frame_groups(1):
            return torch.cat([bn_relu(x[a:b], bn, relu) for a, b in zip(off[:-1], off[1:]) if b > a], 0)
    if bn.training:
        if FUSE_RUNNING and bn.momentum is not None and bn.running_mean.is_contiguous() and \
                bn.running_var.is_contiguous():
            # the running statistics and the batch counter are updated inside the statistics launch of the forward
            z, mean, var = _BNReLUTrain.apply(x, bn.weight, bn.bias, bn.eps, relu, bn.running_mean, bn.running_var,
                                              float(bn.momentum), _counter(bn), slabs)
            return z
        with torch.no_grad():
            bn.num_batches_tracked += 1
        z, mean, var = _BNReLUTrain.apply(x, bn.weight, bn.bias, bn.eps, relu)
        with torch.no_grad():                      # cumulative moving average (momentum=None): factor known on the host
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var, alpha=m * n / (n - 1))
        return z
    if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
        z = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return torch.relu(z) if relu else z
    x = x.contiguous()
    invstd = _invstd(bn)
    z = torch.empty_like(x)
    check(lib.crb_bn_relu_apply(ptr(x), n, C, ptr(bn.running_mean.contiguous()), ptr(invstd), ptr(bn.weight.contiguous()),
                                ptr(bn.bias.contiguous()), int(relu), ptr(z), 0, cur_stream(x.device)), 'crb_bn_relu_apply')
    return z


@torch.no_grad()
def bn_apply_(x, bn, relu=True):
    """in-place inference BatchNorm(+ReLU) on (N, C) rows from the running statistics: one read + one write of x"""
    require_cuda(x)
    n, C = x.shape
    assert x.is_contiguous()
    invstd = _invstd(bn)
    check(lib.crb_bn_relu_apply(ptr(x), n, C, ptr(bn.running_mean.contiguous()), ptr(invstd), ptr(bn.weight.contiguous()),
                                ptr(bn.bias.contiguous()), int(relu), ptr(x), 0, cur_stream(x.device)), 'crb_bn_relu_apply')
    return x


@torch.no_grad()
def bn_apply_into(x, bn, relu, out, col):
    """inference BatchNorm(+ReLU) of (N, C) rows written into columns [col, col+C) of the row-major (N, W) buffer `out`"""
    import ctypes
    require_cuda(x, out)
    n, C = x.shape
    assert x.is_contiguous() and out.is_contiguous() and out.shape[0] == n and col + C <= out.shape[1]
    invstd = _invstd(bn)
    check(lib.crb_bn_relu_apply(ptr(x), n, C, ptr(bn.running_mean.contiguous()), ptr(invstd), ptr(bn.weight.contiguous()),
                                ptr(bn.bias.contiguous()), int(relu), ctypes.c_void_p(out.data_ptr() + 4 * col),
                                out.shape[1], cur_stream(x.device)), 'crb_bn_relu_apply')
    return out


class _BNReLUConcatTrain(torch.autograd.Function):
    """training BatchNorm+ReLU of several (n, C_i) row matrices written side by side into ONE (n, sum C_i) matrix — the
    torch.cat of the BEV backbone's up-sampled branches (base_bev_backbone.py:100-104) without the copy: every branch's
    apply kernel writes its channel slice (row stride sum C_i), the backward reads the slices of the incoming gradient in
    place. Arguments: relu, then (x, gamma, beta, eps, running_mean, running_var, momentum) per branch."""

    @staticmethod
    def forward(ctx, relu, *args):
        import ctypes
        k = len(args) // 7
        xs = [args[7 * i].contiguous() for i in range(k)]
        n, dev = xs[0].shape[0], xs[0].device
        widths = [x.shape[1] for x in xs]
        total = sum(widths)
        out = torch.empty((n, total), dtype=torch.float32, device=dev)
        saved, col = [], 0
        for i, x in enumerate(xs):
            gamma, beta, eps, rm, rv, mom = args[7 * i + 1:7 * i + 7]
            C = widths[i]
            mean = torch.empty((C,), dtype=torch.float32, device=dev)
            var, invstd = torch.empty_like(mean), torch.empty_like(mean)
            wsb = lib.crb_bn_workspace_bytes(n, C)
            ws, tk = _scratch(dev, wsb)
            g, b = gamma.contiguous().float(), beta.contiguous().float()
            zptr = ctypes.c_void_p(out.data_ptr() + 4 * col)
            _bn_check(lib.crb_bn_relu_forward(ptr(x), n, C, ptr(g), ptr(b), float(eps), int(relu), zptr, total, ptr(mean),
                                          ptr(var), ptr(invstd), ptr(rm), ptr(rv), None, float(mom), ptr(ws), wsb, ptr(tk),
                                          cur_stream(dev)), 'crb_bn_relu_forward')
            _touch(rm, rv)
            saved += [x, mean, invstd, g, b]
            col += C
        ctx.save_for_backward(*saved)
        ctx.relu, ctx.widths = int(relu), widths
        return out

    @staticmethod
    def backward(ctx, dz):
        import ctypes
        dz = dz.contiguous().float()
        n, total = dz.shape
        saved = ctx.saved_tensors
        grads, col = [None], 0
        for i, C in enumerate(ctx.widths):
            x, mean, invstd, g, b = saved[5 * i:5 * i + 5]
            dev = x.device
            dx = torch.empty_like(x)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty_like(dgamma)
            wsb = lib.crb_bn_workspace_bytes(n, C)
            ws, tk = _scratch(dev, wsb)
            dzp = ctypes.c_void_p(dz.data_ptr() + 4 * col)
            _bn_check(lib.crb_bn_relu_backward(ptr(x), dzp, total, n, C, ptr(mean), ptr(invstd), ptr(g), ptr(b), ctx.relu,
                                           ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), wsb, ptr(tk), cur_stream(dev)),
                  'crb_bn_relu_backward')
            grads += [dx, dgamma, dbeta, None, None, None, None]
            col += C
        return tuple(grads)


def bn_relu_concat(xs, bns, relu=True):
    """training-mode relu(bn_i(x_i)) for row matrices x_i (n, C_i), concatenated along the channel axis -> (n, sum C_i).
    Every bn must be in training mode with a momentum (running statistics are updated in the forward launch)."""
    if _GROUPS is not None and _GROUPS.G > 1:
        off = _GROUPS.offsets(xs[0])
        if not torch.is_grad_enabled():
            total = sum(x.shape[1] for x in xs)
            out = torch.empty((xs[0].shape[0], total), dtype=torch.float32, device=xs[0].device)
            col = 0
            for x, bn in zip(xs, bns):
                _frames_forward(x, off, bn, relu, out, col)
                col += x.shape[1]
            return out
        with frame_groups(1):
            return torch.cat([bn_relu_concat([x[a:b] for x in xs], bns, relu) for a, b in zip(off[:-1], off[1:])], 0)
    args = []
    for x, bn in zip(xs, bns):
        with torch.no_grad():
            bn.num_batches_tracked += 1
        args += [x, bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, float(bn.momentum)]
    return _BNReLUConcatTrain.apply(relu, *args)


class _BNReLUMaxConcatTrain(torch.autograd.Function):
    """training relu(bn_i(x_i)) followed by the max over groups of ns_i consecutive rows, for several (M*ns_i, C_i) row
    matrices, written side by side into ONE (M, sum C_i) matrix: the BatchNorm2d -> ReLU -> max_pool2d -> torch.cat tail of
    StackSAModuleMSG.forward (pointnet2_modules.py:96-112). The normalised (M*ns_i, C_i) matrices, the zero-filled gradient
    of the max and its scatter never exist. Arguments: (x, ns, gamma, beta, eps, running_mean, running_var, momentum) per
    scale."""

    @staticmethod
    def forward(ctx, *args):
        import ctypes
        k = len(args) // 8
        xs = [args[8 * i].contiguous() for i in range(k)]
        nss = [int(args[8 * i + 1]) for i in range(k)]
        dev = xs[0].device
        M = xs[0].shape[0] // nss[0]
        widths = [x.shape[1] for x in xs]
        total = sum(widths)
        out = torch.empty((M, total), dtype=torch.float32, device=dev)
        saved, col = [], 0
        for i, x in enumerate(xs):
            gamma, beta, eps, rm, rv, mom = args[8 * i + 2:8 * i + 8]
            C, ns = widths[i], nss[i]
            assert x.shape[0] == M * ns
            mean = torch.empty((C,), dtype=torch.float32, device=dev)
            var, invstd = torch.empty_like(mean), torch.empty_like(mean)
            arg = torch.empty((M, C), dtype=torch.int32, device=dev)
            wsb = lib.crb_bn_workspace_bytes(M * ns, C)
            ws, tk = _scratch(dev, wsb)
            g, b = gamma.contiguous().float(), beta.contiguous().float()
            zptr = ctypes.c_void_p(out.data_ptr() + 4 * col)
            _bn_check(lib.crb_bn_relu_max_forward(ptr(x), M, ns, C, ptr(g), ptr(b), float(eps), zptr, total, ptr(arg),
                                              ptr(mean), ptr(var), ptr(invstd), ptr(rm), ptr(rv), None, float(mom), ptr(ws), wsb,
                                              ptr(tk), cur_stream(dev)), 'crb_bn_relu_max_forward')
            _touch(rm, rv)
            saved += [x, mean, invstd, g, b, arg]
            col += C
        ctx.save_for_backward(*saved)
        ctx.widths, ctx.nss = widths, nss
        return out

    @staticmethod
    def backward(ctx, gz):
        import ctypes
        gz = gz.contiguous().float()
        M, total = gz.shape
        saved = ctx.saved_tensors
        grads, col = [], 0
        for i, (C, ns) in enumerate(zip(ctx.widths, ctx.nss)):
            x, mean, invstd, g, b, arg = saved[6 * i:6 * i + 6]
            dev = x.device
            dx = torch.empty_like(x)
            dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
            dbeta = torch.empty_like(dgamma)
            wsb = lib.crb_bn_workspace_bytes(M * ns, C)
            ws, tk = _scratch(dev, wsb)
            gp = ctypes.c_void_p(gz.data_ptr() + 4 * col)
            _bn_check(lib.crb_bn_relu_max_backward(ptr(x), gp, total, ptr(arg), M, ns, C, ptr(mean), ptr(invstd), ptr(g), ptr(b),
                                               ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), wsb, ptr(tk), cur_stream(dev)),
                  'crb_bn_relu_max_backward')
            grads += [dx, None, dgamma, dbeta, None, None, None, None]
            col += C
        return tuple(grads)


def bn_relu_max_concat(xs, nss, bns):
    """training-mode max over groups of nss[i] rows of relu(bn_i(x_i)), concatenated along the channel axis -> (M, sum C_i);
    every bn in training mode with a momentum (running statistics are updated in the forward launch)"""
    if _GROUPS is not None and _GROUPS.G > 1:
        G = _GROUPS.G
        M = xs[0].shape[0] // int(nss[0])
        assert M % G == 0, 'query points must come in equal numbers per frame'
        m = M // G
        if not torch.is_grad_enabled():
            import ctypes
            total = sum(x.shape[1] for x in xs)
            out = torch.empty((M, total), dtype=torch.float32, device=xs[0].device)
            col = 0
            for x, ns, bn in zip(xs, nss, bns):
                x = x.contiguous()
                C = x.shape[1]
                arg = torch.empty((M, C), dtype=torch.int32, device=x.device)
                wsb = lib.crb_bn_frames_workspace_bytes(G, m * int(ns), C)
                ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
                with torch.no_grad():
                    bn.num_batches_tracked += G
                check(lib.crb_bn_relu_max_forward_frames(ptr(x), G, m, int(ns), C, ptr(bn.weight.contiguous().float()),
                                                         ptr(bn.bias.contiguous().float()), float(bn.eps),
                                                         ctypes.c_void_p(out.data_ptr() + 4 * col), total, ptr(arg),
                                                         ptr(bn.running_mean), ptr(bn.running_var), float(bn.momentum),
                                                         ptr(ws), wsb, cur_stream(x.device)),
                      'crb_bn_relu_max_forward_frames')
                _touch(bn.running_mean, bn.running_var)
                col += C
            return out
        with frame_groups(1):
            return torch.cat([bn_relu_max_concat([x[g * m * int(ns):(g + 1) * m * int(ns)] for x, ns in zip(xs, nss)], nss, bns)
                              for g in range(G)], 0)
    args = []
    for x, ns, bn in zip(xs, nss, bns):
        with torch.no_grad():
            bn.num_batches_tracked += 1
        args += [x, int(ns), bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, float(bn.momentum)]
    return _BNReLUMaxConcatTrain.apply(*args)
