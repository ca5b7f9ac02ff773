"""GPU: the split-bf16 Winograd kernel's third form (workgroup tile 32 tiles x 128 channels, crb_conv3x3_winograd4c_nhwc) against the
64 x 64 form: bit-equality of the outputs (bias + ReLU, statistics variant, input gradient) on ragged and bench shapes, then
interleaved timing. Usage: python tools/time_wino4_c.py"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'crb-active-3ddet_amd'))
from crbhip import winograd  # noqa: E402

dev = torch.device('cuda:0')


def images(w):
    out = {}
    for c in (False, True):
        winograd.FORM_C = c
        out[c] = (winograd.weights_forward4(w), winograd.weights_input_grad4(w) if winograd.supported4(w.shape[0], w.shape[1], 31, 1) else None)
    winograd.FORM_C = True
    assert not getattr(out[False][0], '_crb_c', False) and out[True][0]._crb_c
    return out


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


ok_all = True
for (N, C, K, H, W) in [(2, 128, 128, 50, 44), (1, 256, 256, 33, 22), (3, 64, 128, 31, 9), (2, 16, 384, 33, 17), (1, 256, 128, 40, 31),
                        (16, 16, 128, 50, 44), (5, 32, 128, 37, 5), (3, 48, 128, 63, 70), (2, 128, 256, 47, 35), (1, 128, 128, 200, 176)]:
    torch.manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    b = torch.randn(K, device=dev)
    im = images(w)
    ya = winograd.conv3x3_U4(x, im[False][0], b, relu=True)
    yc = winograd.conv3x3_U4(x, im[True][0], b, relu=True)
    sa = winograd.conv3x3_stats_U4(x, im[False][0])
    sc = winograd.conv3x3_stats_U4(x, im[True][0])
    eq = torch.equal(ya, yc) and torch.equal(sa[0], sc[0])
    st = float((sa[1].double().sum(0) - sc[1].double().sum(0)).abs().max() / sa[1].double().sum(0).abs().max())
    dg = True
    if im[True][1] is not None and winograd.supported4(K, C, H, W) and C % 128 == 0:
        dy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        dg = torch.equal(winograd.conv3x3_U4(dy, im[False][1]), winograd.conv3x3_U4(dy, im[True][1]))
    good = eq and st < 1e-5 and dg
    ok_all = ok_all and good
    print('%d x %d -> %d @ %d x %d: outputs bit-equal %s, statistics (other slabs) %.1e, input gradient bit-equal %s' % (N, C, K, H, W, eq, st, dg), flush=True)
print('ALL OK' if ok_all else 'SOME FAILED', flush=True)

for (N, C, K, H, W) in [(16, 128, 128, 200, 176), (16, 256, 128, 200, 176), (16, 256, 256, 100, 88), (16, 128, 256, 100, 88)]:
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    im = images(w)
    print('%d x %d -> %d @ %d x %d: bit-equal %s' % (N, C, K, H, W, torch.equal(winograd.conv3x3_U4(x, im[False][0]), winograd.conv3x3_U4(x, im[True][0]))), flush=True)
    for rep in range(3):
        for c, name in ((False, '64 tiles x 64 channels'), (True, '32 tiles x 128 channels')):
            t = timeit(lambda: winograd.conv3x3_U4(x, im[c][0]))
            print('%d x %d -> %d @ %d x %d  %-24s %.1f us' % (N, C, K, H, W, name, t), flush=True)
