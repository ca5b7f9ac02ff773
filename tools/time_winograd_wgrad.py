#!/usr/bin/env python
"""Winograd-domain weight gradient (csrc/winograd_wgrad.hip) against MIOpen's wrw on the stride-1 3x3 shapes of the BEV backbone:
error against an f64 weight gradient, run-to-run bit equality, time per call, skip-work builds.
usage: python tools/time_winograd_wgrad.py"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timeit(fn, it=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def wgrad_miopen(x, dy, w):
    return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


if __name__ == '__main__':
    from crbhip import winograd, lib
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    small = ((2, 64, 64, 9, 11), (1, 128, 64, 40, 31), (3, 64, 192, 7, 5), (2, 128, 128, 37, 29))
    big = ((16, 128, 128, 200, 176), (16, 256, 256, 100, 88), (16, 256, 128, 200, 176))
    for (N, C, K, H, W) in small + big:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        ref = wgrad_miopen(x, dy, w)
        got = winograd.conv3x3_wgrad(x, dy, w)
        again = winograd.conv3x3_wgrad(x, dy, w)
        got_c = winograd.conv3x3_wgrad(x, dy, w.contiguous())
        if N * H * W <= 40000:
            x64, w64 = x.double(), w.double().requires_grad_(True)
            F.conv2d(x64, w64, None, padding=1).backward(dy.double())
            r64 = w64.grad
        else:
            r64 = ref.double()
        sc = float(r64.abs().max())
        print('%dx%d->%d @%dx%d: weight gradient error / largest entry: winograd %.2e, MIOpen %.2e (reference: %s); bit-equal rerun %s; '
              'contiguous weight layout equal %s' % (N, C, K, H, W, float((got.double() - r64).abs().max()) / sc,
                                                    float((ref.double() - r64).abs().max()) / sc, 'f64' if N * H * W <= 40000 else 'MIOpen',
                                                    bool(torch.equal(got, again)), bool(torch.equal(got_c, got))), flush=True)
        if (N, C, K, H, W) not in big:
            continue
        flops = 2.0 * N * H * W * 9 * C * K
        t_m, _ = timeit(lambda: wgrad_miopen(x, dy, w))
        t_w, t_w_min = timeit(lambda: winograd.conv3x3_wgrad(x, dy, w))
        tm = []
        for mode in (1, 2, 3):
            lib.crb_winograd2_wgrad_set_mode(mode)
            tm.append(timeit(lambda: winograd.conv3x3_wgrad(x, dy, w), it=10, warm=3)[0])
        lib.crb_winograd2_wgrad_set_mode(0)
        print('   weight gradient: MIOpen %.0f us (%.0f TF direct-equivalent) | winograd %.0f us (min %.0f; %.0f TF direct-equivalent, %.0f TF of MFMA work)'
              ' | skip-work builds: no MFMAs %.0f us, no transforms %.0f us, no loads %.0f us'
              % (t_m, flops / t_m / 1e6, t_w, t_w_min, flops / t_w / 1e6, flops / 2.25 / t_w / 1e6, tm[0], tm[1], tm[2]), flush=True)
