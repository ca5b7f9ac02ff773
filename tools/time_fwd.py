#!/usr/bin/env python
"""Stable timing of the gather-GEMM forward on the SECOND bs=16 subm tables (levels 1..4): 300 warm-up launches (clocks),
then 12 x 40 launches, median and min per level — single short measurements on this pool differ by 10-15 % between runs.
usage: python tools/time_fwd.py [f32|bf16x3]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import sparse, voxel
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    ARITH = sys.argv[1] if len(sys.argv) > 1 else 'f32'
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
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        n, c = rb.n_out, chans[lvl]
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27, c, c, device=dev) / 10
        table = rb.table_for('nbr', c, c, ARITH)
        P = table.num_pairs() if hasattr(table, 'num_pairs') else int((rb.nbr >= 0).sum())
        for _ in range(300):
            sparse._conv_forward_raw(x, w, table, n, arithmetic=ARITH)
        torch.cuda.synchronize()
        ts = []
        for rep in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                sparse._conv_forward_raw(x, w, table, n, arithmetic=ARITH)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
        t = float(np.median(ts))
        balg = 4.0 * n * c * 2 + 8.0 * P + 4.0 * 27 * c * c
        print('%s L%d subm %dx%d N=%d P=%d: median %.1f us (min %.1f) = %.1f TF, %.0f GB/s alg' % (
            ARITH, lvl, c, c, n, P, t, min(ts), 2.0 * P * c * c / t / 1e6, balg / t / 1e3), flush=True)
