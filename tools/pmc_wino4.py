#!/usr/bin/env python
"""Minimal driver for PMC passes over the split-bf16 Winograd kernel: 16 x 128 -> 128 @ 200 x 176 (or argv: N C K H W), 3 launches.
CRB_WINO4_VARIANT=1 (with CRB_MEASURE_LIB=1): the first form."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import winograd, lib
    N, C, K, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else (16, 128, 128, 200, 176)
    if os.environ.get('CRB_WINO4_VARIANT'):
        lib.crb_winograd4_set_variant(int(os.environ['CRB_WINO4_VARIANT']))
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
    U4 = winograd.weights_forward4(w)
    for _ in range(3):
        winograd.conv3x3_U4(x, U4)
    torch.cuda.synchronize()
    print('PMC_WINO4 %dx%d->%d @%dx%d' % (N, C, K, H, W))
