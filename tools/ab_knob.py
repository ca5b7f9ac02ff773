#!/usr/bin/env python
"""Interleaved A/B of one integer knob of the f32 gather-GEMM forward (C-ABI setter name on the command line) on the SECOND
bs=16 subm tables: 300 warm-up launches, 10 x 40 launches per arm, median; results must be bit-equal.
usage: python tools/ab_knob.py crb_sparse_conv_set_single_w [levels]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import sparse, voxel, lib
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    knob = getattr(lib, sys.argv[1])
    levels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '2,3,4').split(',')]
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    chans = {1: 16, 2: 32, 3: 64, 4: 64}
    for lvl in range(1, 5):
        if lvl > 1:
            rbs = sparse.spconv_rulebook(coords, shape, 16, *geo[lvl - 2])
            coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        if lvl not in levels:
            continue
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        n, c = rb.n_out, chans[lvl]
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27, c, c, device=dev) / 10
        table = rb.table_for('nbr', c, c)
        for _ in range(300):
            sparse._conv_forward_raw(x, w, table, n)
        torch.cuda.synchronize()
        res, ys = {0: [], 1: []}, {}
        for rep in range(10):
            for arm in (0, 1):
                knob(arm)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    ys[arm] = sparse._conv_forward_raw(x, w, table, n)
                e1.record()
                torch.cuda.synchronize()
                res[arm].append(e0.elapsed_time(e1) / 40 * 1e3)
        knob(0)
        print('L%d %dx%d: knob 0 median %.1f us (min %.1f) | knob 1 median %.1f us (min %.1f) | equal %s' % (
            lvl, c, c, np.median(res[0]), min(res[0]), np.median(res[1]), min(res[1]), bool(torch.equal(ys[0], ys[1]))),
            flush=True)
