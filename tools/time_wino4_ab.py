"""GPU, measurement library: the split-bf16 Winograd product kernel with its weight fragments loaded straight into registers
(variant 1, the product form) against the same kernel with the weight images copied through LDS (variant 3), interleaved in one
process at the bench shapes. Usage: python tools/time_wino4_ab.py"""
import os
import sys
import numpy as np
import torch

os.environ['CRB_MEASURE_LIB'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'crb-active-3ddet_amd'))
from crbhip import winograd, lib  # noqa: E402

dev = torch.device('cuda:0')
NAMES = {1: 'U in registers', 3: 'U through LDS-DMA'}
VARIANTS = (1, 3)


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


for (N, C, K, H, W) in [(2, 128, 128, 50, 44), (1, 256, 256, 33, 22), (3, 64, 64, 31, 9), (2, 16, 192, 33, 17), (1, 256, 128, 40, 31),
                        (16, 16, 64, 50, 44), (5, 32, 64, 37, 5), (3, 48, 128, 63, 70)]:
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    b = torch.randn(K, device=dev)
    U4 = winograd.weights_forward4(w)
    out, st = {}, {}
    for v in VARIANTS:
        lib.crb_winograd4_set_variant(v)
        out[v] = winograd.conv3x3_U4(x, U4, b, relu=True)
        st[v] = winograd.conv3x3_stats_U4(x, U4)
    print('%d x %d -> %d @ %d x %d: bit-equal outputs %s, statistics %s' % (N, C, K, H, W, all(torch.equal(out[1], out[v]) for v in VARIANTS),
          all(torch.equal(st[1][0], st[v][0]) and torch.equal(st[1][1], st[v][1]) for v in VARIANTS)), flush=True)
lib.crb_winograd4_set_variant(1)

for (N, C, K, H, W) in [(16, 128, 128, 200, 176), (16, 256, 128, 200, 176), (16, 256, 256, 100, 88), (16, 128, 256, 100, 88)]:
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    U4 = winograd.weights_forward4(w)
    out = {}
    for v in VARIANTS:
        lib.crb_winograd4_set_variant(v)
        out[v] = winograd.conv3x3_U4(x, U4)
    print('%d x %d -> %d @ %d x %d: outputs of the forms bit-equal: %s' % (N, C, K, H, W, all(torch.equal(out[1], out[v]) for v in VARIANTS)), flush=True)
    for rep in range(3):
        for v in VARIANTS:
            lib.crb_winograd4_set_variant(v)
            t = timeit(lambda: winograd.conv3x3_U4(x, U4))
            print('%d x %d -> %d @ %d x %d  %-20s %.1f us' % (N, C, K, H, W, NAMES[v], t), flush=True)
    lib.crb_winograd4_set_variant(1)
