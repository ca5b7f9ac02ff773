#!/usr/bin/env python
"""the second Winograd F(2x2,3x3) kernel (csrc/winograd_conv2.hip) against MIOpen's implicit GEMM and the
first kernel on the stride-1 3x3 shapes of the BEV backbone — error against an f64 convolution on unit-scale data, run-to-run
bit equality, time per call and the skip-work measurement builds.  usage: python tools/time_winograd2.py [--quick]"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timeit(fn, it=30, warm=8):
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


if __name__ == '__main__':
    from crbhip import winograd, lib
    quick = '--quick' in sys.argv
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    small = ((2, 128, 128, 37, 29), (3, 64, 64, 33, 17), (5, 32, 192, 31, 9), (2, 128, 64, 40, 31), (3, 64, 64, 7, 5), (2, 8, 64, 9, 11), (1, 256, 128, 40, 31), (20, 24, 64, 5, 9), (7, 16, 128, 6, 4))
    big = ((16, 128, 128, 200, 176), (16, 256, 256, 100, 88), (16, 256, 128, 200, 176))
    for (N, C, K, H, W) in small + big:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
        b = torch.randn(K, device=dev)
        U2 = winograd.weights_forward2(w)
        y = winograd.conv3x3_U2(x, U2, b)
        y_again = winograd.conv3x3_U2(x, U2, b)
        ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        scale = float(ref64.abs().max())
        e_w = float((y.double() - ref64).abs().max()) / scale
        ref = F.conv2d(x, w, b, padding=1)
        e_m = float((ref.double() - ref64).abs().max()) / scale
        yr = winograd.conv3x3_U2(x, U2, b, relu=True)
        dy = torch.randn_like(ref)
        dx_ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
        e_dx = None
        if winograd.supported2(K, C, H, W):
            dx = winograd.conv3x3_U2(dy, winograd.weights_input_grad2(w))
            e_dx = float((dx.double() - dx_ref).abs().max() / dx_ref.abs().max())
        print('%dx%d->%d @%dx%d: error vs f64 conv / output scale: winograd2 %.2e, MIOpen %.2e; input grad %s; bit-equal rerun %s; relu ok %s'
              % (N, C, K, H, W, e_w, e_m, 'n/a' if e_dx is None else '%.2e' % e_dx, bool(torch.equal(y, y_again)),
                 bool(torch.equal(yr, torch.relu(y)))), flush=True)
        if (N, C, K, H, W) not in big:
            continue
        flops = 2.0 * N * H * W * 9 * C * K
        t_m, t_m_min = timeit(lambda: F.conv2d(x, w, b, padding=1))
        t_2, t_2_min = timeit(lambda: winograd.conv3x3_U2(x, U2, b))
        t_u, _ = timeit(lambda: winograd.weights_forward2(w))
        line = '   forward: MIOpen %.0f us (min %.0f; %.0f TF direct-equivalent) | winograd2 %.0f us (min %.0f; %.0f TF direct-equivalent, %.0f TF of MFMA work) + weight image %.0f us' % (
            t_m, t_m_min, flops / t_m / 1e6, t_2, t_2_min, flops / t_2 / 1e6, flops / 2.25 / t_2 / 1e6, t_u)
        if winograd.supported(C, K):
            U1 = winograd.weights_forward(w)
            t_1, _ = timeit(lambda: winograd.conv3x3_U(x, U1, b))
            line += ' | first kernel %.0f us' % t_1
        print(line, flush=True)
        if not quick:
            tm = []
            for mode in (1, 2, 3):
                lib.crb_winograd2_set_mode(mode)
                tm.append(timeit(lambda: winograd.conv3x3_U2(x, U2, b))[0])
            lib.crb_winograd2_set_mode(0)
            print('   measurement builds: no MFMAs %.0f us, no transform %.0f us, no DMA in the loop %.0f us' % tuple(tm), flush=True)
