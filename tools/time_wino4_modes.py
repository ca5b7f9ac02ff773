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
NAMES = {0: 'product', 1: 'no MFMAs', 2: 'no transform', 3: 'no DMA in the loop', 4: 'no operand reads', 5: 'no U copies', 6: 'no raw copies',
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
    for rep in range(2):
        for mode in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            lib.crb_winograd4_set_mode(mode)
            t = timeit(lambda: winograd.conv3x3_U4(x, U4))
            print('%d x %d -> %d @ %d x %d  mode %d (%s): %.1f us' % (N, C, K, H, W, mode, NAMES[mode], t), flush=True)
        lib.crb_winograd4_set_mode(0)

# mode 9: where a wave's time goes (s_memtime sums per wave, second form)
N, C, K, H, W = 16, 128, 128, 200, 176
x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
U4 = winograd.weights_forward4(w)
dbg = torch.zeros((256 * 8 * 8,), dtype=torch.int64, device=dev)
lib.crb_winograd4_set_debug(dbg.data_ptr())
lib.crb_winograd4_set_mode(9)
for _ in range(3):
    winograd.conv3x3_U4(x, U4)
torch.cuda.synchronize()
t9 = timeit(lambda: winograd.conv3x3_U4(x, U4))
lib.crb_winograd4_set_mode(0)
lib.crb_winograd4_set_debug(None)
d = dbg.cpu().numpy().reshape(256, 8, 8).astype(np.float64)
print('mode 9 (stamps): %.1f us per launch' % t9)
names = ['counter wait', 'barrier', 'phase head', 'phase body', 'epilogue', 'total']
for wv in range(8):
    phases = d[:, wv, 6] * 4
    print('wave %d: ' % wv + ', '.join('%s %.0f' % (n, (d[:, wv, k] / (phases if k < 4 else d[:, wv, 7] if k == 4 else phases)).mean())
                                        for k, n in enumerate(names)) + '  (cycles per phase; epilogue per unit; total per phase)')
