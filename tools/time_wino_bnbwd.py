#!/usr/bin/env python
"""what the BatchNorm-backward sums cost in the Winograd input-gradient kernel's epilogue
(crb_conv3x3_winograd2_bnbwd_nhwc) against the plain launch + the reduction pass it replaces (crb_bn_relu_backward with dx = NULL),
and that the slab sums reduce to that pass's dbeta / dgamma. usage: python tools/time_wino_bnbwd.py"""
import os
import sys
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402
from crbhip import lib, check, ptr, cur_stream, winograd, bnrelu  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
st = cur_stream(dev)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return t[len(t) // 2]


for (N, C, H, W) in ((16, 128, 200, 176), (16, 256, 100, 88), (2, 128, 37, 21)):
    dy = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    U = winograd.weights_input_grad2(w)
    yprev = torch.randn(N, H, W, C, device=dev)
    mean, invstd = yprev.view(-1, C).mean(0).contiguous(), (yprev.view(-1, C).var(0, unbiased=False) + 1e-3).rsqrt().contiguous()
    gamma, beta = (torch.randn(C, device=dev) * 0.5 + 0.8).contiguous(), (torch.randn(C, device=dev) * 0.3).contiguous()
    dz = torch.empty(N, H, W, C, device=dev)
    dz2 = torch.empty(N, H, W, C, device=dev)
    nsl = int(lib.crb_winograd2_stats_slabs(N, H, W))
    slabs = torch.empty(nsl, 2, C, device=dev)
    n = N * H * W
    wsb = lib.crb_bn_workspace_bytes(n, C)
    ws, tk = bnrelu._scratch(dev, wsb)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dg2, db2 = torch.empty(C, device=dev), torch.empty(C, device=dev)

    def plain():
        check(lib.crb_conv3x3_winograd2_nhwc(ptr(dy), ptr(U), ptr(dz), N, H, W, C, C, None, 0, st), 'plain')

    def fused():
        check(lib.crb_conv3x3_winograd2_bnbwd_nhwc(ptr(dy), ptr(U), ptr(dz2), ptr(slabs), N, H, W, C, C, ptr(yprev), ptr(mean), ptr(invstd),
                                                   ptr(gamma), ptr(beta), 1, st), 'bnbwd')

    def reduce_pass():
        check(lib.crb_bn_relu_backward(ptr(yprev), ptr(dz), 0, n, C, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), 1, None, ptr(dg), ptr(db),
                                       ptr(ws), wsb, ptr(tk), st), 'bn backward sums')

    def from_slabs():
        check(lib.crb_bn_relu_backward_partials(ptr(yprev), ptr(dz2), 0, n, C, ptr(slabs), nsl, ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                                1, None, ptr(dg2), ptr(db2), ptr(ws), wsb, ptr(tk), st), 'bn backward from slabs')
    plain(); fused(); reduce_pass(); from_slabs()
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print('%dx%dx%dx%d: dz equal %s; dbeta %.1e, dgamma %.1e (relative to the reduction pass)' % (
        N, C, H, W, torch.equal(dz, dz2), rel(db2, db), rel(dg2, dg)))
    tp, tf, tr, ts = timed(plain), timed(fused), timed(reduce_pass), timed(from_slabs)
    print('    plain launch %.1f us, with the epilogue sums %.1f us (+%.1f); reduction pass %.1f us, slab reduction %.1f us: net %.1f us per layer'
          % (tp, tf, tf - tp, tr, ts, (tp + tr) - (tf + ts)))
