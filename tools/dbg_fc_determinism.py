#!/usr/bin/env python
"""GPU: is the RoI head's first FC layer in eval (torch.addmm, (n, 27648) x (27648, 256), hipBLASLt) the source of the 1.6e-8
schedule difference of tests/test_pvrcnn_gpu.py? Same inputs, many calls, the allocator perturbed in between (other live tensors,
other sizes freed): count the distinct output bit patterns, for the vendor call and for the K-sliced batched form."""
import os
import sys
import hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    for n in (200, 1600):
        x = torch.randn(n, 27648, device=dev)
        w = torch.randn(256, 27648, device=dev) / 166.0
        b = torch.randn(256, device=dev)
        pats = {'addmm': set(), 'sliced': set(), 'addmm_moved': set()}
        keep = []
        for rep in range(60):
            if rep % 3 == 0:
                keep.append(torch.empty((1 + 37 * rep) * 1024, device=dev))
            if rep % 7 == 0 and keep:
                keep.pop(0)
                torch.cuda.empty_cache()
            y = torch.addmm(b, x, w.t())
            pats['addmm'].add(hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest())
            x2 = x.clone()                      # the same values at another address
            y2 = torch.addmm(b, x2, w.t())
            pats['addmm_moved'].add(hashlib.sha1(y2.cpu().numpy().tobytes()).hexdigest())
            ys = torch.bmm(x2.view(n, 216, 128).transpose(0, 1), w.view(256, 216, 128).permute(1, 2, 0)).sum(0) + b
            pats['sliced'].add(hashlib.sha1(ys.cpu().numpy().tobytes()).hexdigest())
        print('n = %5d: distinct outputs over 60 calls: addmm %d, addmm on a moved copy %d, union %d; K-sliced bmm + sum %d' %
              (n, len(pats['addmm']), len(pats['addmm_moved']), len(pats['addmm'] | pats['addmm_moved']), len(pats['sliced'])), flush=True)
