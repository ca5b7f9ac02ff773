#!/usr/bin/env python
"""f32 gather-GEMM forward at 64x64 on the SECOND bs=16 level-3 / level-4 tables, (N,K)-table kernels: v2 (one 16-row tile
per wave, 64-row workgroups) against v1 with 1 / 2 / 4 row tiles per wave sharing every W[o] fragment (64 / 128 / 256-row
workgroups). Interleaved, 8 x 40 launches per arm, median."""
import os, sys
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch

if __name__ == '__main__':
    from crbhip import sparse, voxel, lib
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    for lvl in range(1, 5):
        if lvl > 1:
            rbs = sparse.spconv_rulebook(coords, shape, 16, *geo[lvl - 2])
            coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        if lvl < 3:
            continue
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        n, c = rb.n_out, 64
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27, c, c, device=dev) / 10
        table = rb.sorted_table('nbr')
        compact = rb.compact_table('nbr')
        arms = {8: 'v2, 1 tile/wave', 1: 'v1, 1 tile/wave', 2: 'v1, 2 tiles/wave', 4: 'v1, 4 tiles/wave'}
        for _ in range(200):
            sparse._conv_forward_raw(x, w, table, n)
        res = {k: [] for k in list(arms) + ['compact']}
        for rep in range(8):
            for k in res:
                lib.crb_sparse_conv_set_subtiles(0 if k == 'compact' else k)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    sparse._conv_forward_raw(x, w, compact if k == 'compact' else table, n)
                e1.record()
                torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1) / 40 * 1e3)
        lib.crb_sparse_conv_set_subtiles(0)
        print('L%d 64x64: compact-table v2 (product) %.1f us | (N,K)-table kernels: ' % (lvl, np.median(res['compact'])) +
              ' | '.join('%s %.1f us' % (arms[k], np.median(res[k])) for k in arms), flush=True)
