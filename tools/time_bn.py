#!/usr/bin/env python
"""BatchNorm+ReLU passes on the BEV-sized row matrices, with a producer in front of each (a copy that writes the input in
ascending order, as the convolution before it does), for the four traversal orders of the measurement knob crb_bn_set_order
(bit 0: statistics passes walk the rows from the end, bit 1: apply passes do). Answers how much of the second read of a tensor
the 256 MB Infinity Cache serves. Usage: python tools/time_bn.py"""
import os, sys
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch

if __name__ == '__main__':
    from crbhip import lib, bnrelu
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    for (n, C) in ((563200, 128), (140800, 256), (563200, 256)):
        src = torch.randn(n, C, device=dev)
        src2 = torch.randn(n, C, device=dev)
        x = torch.empty_like(src).requires_grad_(True)
        dz = torch.empty_like(src)
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
        mb = n * C * 4 / 1e6
        res = {}
        for rounds in range(2):                 # interleaved: every order measured in both rounds
            for order in (0, 1, 2, 3):
                lib.crb_bn_set_order(order)
                tf, tb = [], []
                for it in range(12):
                    with torch.no_grad():
                        x.copy_(src)                                   # producer of x
                    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                    e[0].record()
                    z = bnrelu.bn_relu(x, bn, True)
                    e[1].record()
                    dz.copy_(src2)                                     # producer of dz
                    e[2].record()
                    z.backward(dz)
                    e[3].record()
                    torch.cuda.synchronize()
                    x.grad = None
                    if it >= 2:
                        tf.append(e[0].elapsed_time(e[1]) * 1e3)
                        tb.append(e[2].elapsed_time(e[3]) * 1e3)
                res.setdefault(order, []).append((float(np.median(tf)), float(np.median(tb))))
        for order in (0, 1, 2, 3):
            f = np.mean([r[0] for r in res[order]]); b = np.mean([r[1] for r in res[order]])
            print('%7d x %3d (%5.0f MB)  order %d (stats %s, apply %s): fwd %6.1f us = %4.2f TB/s of 3 passes, bwd %6.1f us = %4.2f TB/s '
                  'of 5 passes' % (n, C, mb, order, 'desc' if order & 1 else 'asc ', 'desc' if order & 2 else 'asc ', f,
                                   3 * mb / f, b, 5 * mb / b))
