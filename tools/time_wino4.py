"""GPU: correctness of crb_conv3x3_winograd4_nhwc (split-bf16 Winograd) against f64 convolutions next to the f32-MFMA kernel, then
same-process interleaved timing of both on the bench shapes. Usage: python tools/time_wino4.py [--quick]"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'crb-active-3ddet_amd'))
from crbhip import winograd  # noqa: E402

dev = torch.device('cuda:0')


def check(N, C, K, H, W, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    b = torch.randn(K, device=dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale = float(ref.abs().max())
    U4 = winograd.weights_forward4(w)
    y4 = winograd.conv3x3_U4(x, U4, b)
    e4 = float((y4.double() - ref).abs().max()) / scale
    r4 = float((y4.double() - ref).norm() / ref.norm())
    U2 = winograd.weights_forward2(w)
    y2 = winograd.conv3x3_U2(x, U2, b)
    e2 = float((y2.double() - ref).abs().max()) / scale
    r2 = float((y2.double() - ref).norm() / ref.norm())
    same = torch.equal(winograd.conv3x3_U4(x, U4, b), y4)
    relu_ok = torch.equal(winograd.conv3x3_U4(x, U4, b, relu=True), torch.relu(y4))
    # statistics variant
    ys, st = winograd.conv3x3_stats_U4(x, U4)
    y0 = winograd.conv3x3_U4(x, U4)
    st_ok = torch.equal(ys, y0)
    s1 = float((st.double().sum(0)[0] - y0.double().sum((0, 2, 3))).abs().max() / (y0.double() ** 2).sum((0, 2, 3)).sqrt().max())
    s2 = float((st.double().sum(0)[1] - (y0.double() ** 2).sum((0, 2, 3))).abs().max() / (y0.double() ** 2).sum((0, 2, 3)).max())
    # input gradient
    dy = torch.randn_like(y4)
    want = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    eg = -1.0
    if winograd.supported4(K, C, H, W):
        dx = winograd.conv3x3_U4(dy, winograd.weights_input_grad4(w))
        eg = float((dx.double() - want).abs().max()) / float(want.abs().max())
    print('N%d C%d K%d %dx%d: wino4 max %.2e rms %.2e | wino2 max %.2e rms %.2e | rerun equal %s relu %s stats-y equal %s s1 %.1e s2 %.1e | dgrad %.2e'
          % (N, C, K, H, W, e4, r4, e2, r2, same, relu_ok, st_ok, s1, s2, eg), flush=True)
    return e4 <= 1e-5 and same and relu_ok and st_ok and s1 < 1e-5 and s2 < 1e-5 and eg <= 2e-5


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    ok = True
    shapes = [(2, 128, 128, 50, 44), (1, 256, 256, 33, 22), (3, 64, 64, 31, 9), (2, 16, 192, 33, 17), (1, 256, 128, 40, 31),
              (16, 16, 64, 50, 44), (5, 32, 64, 37, 5), (3, 48, 128, 63, 70)]
    for s in shapes:
        ok = check(*s) and ok
    print('ALL OK' if ok else 'SOME FAILED', flush=True)
    if '--quick' in sys.argv:
        return
    for (N, C, K, H, W) in [(16, 128, 128, 200, 176), (16, 256, 128, 200, 176), (16, 256, 256, 100, 88), (16, 128, 256, 100, 88)]:
        x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
        U4 = winograd.weights_forward4(w)
        U2 = winograd.weights_forward2(w)
        ok = check(N, C, K, H, W) and ok
        for rep in range(3):
            t2 = timeit(lambda: winograd.conv3x3_U2(x, U2))
            t4 = timeit(lambda: winograd.conv3x3_U4(x, U4))
            gf = 2 * N * ((H + 1) // 2) * ((W + 1) // 2) * C * K * 16 / 1e9
            print('%d x %d -> %d @ %d x %d: wino2 %.1f us (%.1f TF f32 MFMA), wino4 %.1f us (%.1f TF of bf16 MFMA issued, x%.2f)'
                  % (N, C, K, H, W, t2, gf / t2 * 1e-3 * 1e3, t4, 6 * gf / t4 * 1e-3 * 1e3, t2 / t4), flush=True)
    tw = timeit(lambda: winograd.weights_forward4(w), 50)
    print('weight image (256 -> 256): %.1f us' % tw)
    print('ALL OK' if ok else 'SOME FAILED', flush=True)


if __name__ == '__main__':
    main()
