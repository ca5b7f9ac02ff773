"""GPU, measurement library: skip-work builds of the split-bf16 Winograd kernel (crb_winograd4_set_mode) at the bench shapes.
Usage: CRB_MEASURE_LIB=1 python tools/time_wino4_modes.py"""
import os
import sys
import numpy as np
import torch

os.environ['CRB_MEASURE_LIB'] = '1'
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'crb-active-3ddet_amd'))
from crbhip import winograd, lib  # noqa: E402

dev = torch.device('cuda:0')
NAMES = {0: 'product', 1: 'no MFMAs', 2: 'no transform', 3: 'no DMA / U loads in the loop', 4: 'no LDS operand reads', 5: 'no U loads', 6: 'no raw copies',
         7: 'no V stores', 8: 'no output stores'}


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


for (N, C, K, H, W) in [(16, 128, 128, 200, 176), (16, 256, 256, 100, 88)]:
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    U4 = winograd.weights_forward4(w)
    for rep in range(1):
        for mode in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            lib.crb_winograd4_set_mode(mode)
            t = timeit(lambda: winograd.conv3x3_U4(x, U4))
            print('%d x %d -> %d @ %d x %d  mode %d (%s): %.1f us' % (N, C, K, H, W, mode, NAMES[mode], t), flush=True)
        lib.crb_winograd4_set_mode(0)

if '--second-form' not in sys.argv:
    sys.exit(0)
# second form: what each ingredient costs beside the MFMAs alone
NAMES2 = {0: 'product', 64 + 31: 'counters + barriers only', 64 + 30: 'MFMAs only', 64 + 29: 'transform only', 64 + 28: 'MFMAs + transform',
          64 + 22: 'MFMAs + operand reads', 64 + 26: 'MFMAs + copies', 64 + 14: 'MFMAs + output stores', 64 + 20: 'MFMAs + transform + operand reads',
          1: 'all but the MFMAs'}
N, C, K, H, W = 16, 128, 128, 200, 176
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
U4 = winograd.weights_forward4(w)
for rep in range(2):
    for mode, name in NAMES2.items():
        lib.crb_winograd4_set_mode(mode)
        t = timeit(lambda: winograd.conv3x3_U4(x, U4))
        print('%d x %d -> %d @ %d x %d  %-36s %.1f us' % (N, C, K, H, W, name, t), flush=True)
    lib.crb_winograd4_set_mode(0)
