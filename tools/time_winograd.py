#!/usr/bin/env python
"""VERDICT r02 item 9 (stretch, with a kill criterion): hand-written F(2x2,3x3) Winograd f32 convolution against MIOpen's
implicit GEMM on the two stride-1 3x3 shapes of the BEV backbone — error against F.conv2d on unit-scale data and time per
call (forward, and the input gradient as the same kernel on dy).  usage: python tools/time_winograd.py"""
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
    return float(np.median(ts))


if __name__ == '__main__':
    from crbhip import winograd
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    for (N, C, K, H, W) in ((16, 128, 128, 200, 176), (16, 256, 256, 100, 88), (2, 128, 128, 37, 29)):
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
        b = torch.randn(K, device=dev)
        ref = F.conv2d(x, w, b, padding=1)
        U = winograd.weights_forward(w)
        y = winograd.conv3x3_U(x, U, b)
        ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        scale = float(ref64.abs().max())
        e_w = float((y.double() - ref64).abs().max()) / scale
        e_m = float((ref.double() - ref64).abs().max()) / scale
        dy = torch.randn_like(ref)
        dx_ref = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        dx = winograd.conv3x3_U(dy, winograd.weights_input_grad(w))
        e_dx = float((dx - dx_ref).abs().max() / dx_ref.abs().max())
        yr = winograd.conv3x3_U(x, U, b, relu=True)
        assert torch.equal(yr, torch.relu(y))
        flops = 2.0 * N * H * W * 9 * C * K
        t_m = timeit(lambda: F.conv2d(x, w, b, padding=1))
        t_w = timeit(lambda: winograd.conv3x3_U(x, U, b))
        t_u = timeit(lambda: winograd.weights_forward(w))
        t_mb = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
        from crbhip import lib
        tm = []
        for mode in (1, 2):
            lib.crb_winograd_set_mode(mode)
            tm.append(timeit(lambda: winograd.conv3x3_U(x, U, b)))
        lib.crb_winograd_set_mode(0)
        print('   measurement builds: no MFMAs %.0f us, no staging %.0f us' % tuple(tm))
        print('%dx%d->%d @%dx%d: error vs f64 conv / output scale: winograd %.2e, MIOpen %.2e; input grad vs MIOpen %.2e' % (N, C, K, H, W, e_w, e_m, e_dx))
        print('   forward: MIOpen %.0f us (%.0f TF direct-equivalent) | winograd %.0f us (%.0f TF direct-equivalent, %.0f TF of MFMA work) '
              '+ weight transform %.0f us | input grad: MIOpen %.0f us' % (t_m, flops / t_m / 1e6, t_w, flops / t_w / 1e6, flops / 2.25 / t_w / 1e6, t_u, t_mb),
              flush=True)
