#!/usr/bin/env python
"""A/B of the row sort key of the gather-GEMM tables (crb_mask_sort_set_rank_bits): numeric mask order vs rarest-offset-first,
on the SECOND bs=16 subm tables. Tables are rebuilt per arm; forward timed with the stable protocol (300 warm-up launches,
10 x 40 launches, median), arms interleaved; MFMA tile fill = useful / issued 16-row MFMA passes computed from the tables."""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def tile_fill(table, n):
    """pairs / (16 x sum over 16-row tiles of the offsets present in the tile)"""
    m = table.cmask.long()
    pad = (-n) % 16
    if pad:
        m = torch.cat([m, m.new_zeros(pad)])
    m = m.view(-1, 16)
    union = torch.zeros(m.shape[0], dtype=torch.long, device=m.device)
    for k in range(16):
        union |= m[:, k]
    issued = sum(int(((union >> b) & 1).sum()) for b in range(27)) * 16
    return table.num_pairs() / issued


if __name__ == '__main__':
    from crbhip import sparse, voxel, lib
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
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
        c = chans[lvl]
        tabs, ys = {}, {}
        for arm in (0, 1, 2, 3, 4):                  # 3 / 4: per-chunk ranking with 8192- / 16384-row chunks (radix path)
            lib.crb_mask_sort_set_rank_bits(min(arm, 2))
            rows = {3: 8192, 4: 16384}.get(arm, 4096)
            sparse.MASK_SORT_CHUNK = rows
            rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
            tabs[arm] = (rb, rb.table_for('nbr', c, c))
        lib.crb_mask_sort_set_rank_bits(2)
        sparse.MASK_SORT_CHUNK = 4096
        n = tabs[0][0].n_out
        x = torch.randn(n, c, device=dev)
        w = torch.randn(27, c, c, device=dev) / 10
        for _ in range(300):
            sparse._conv_forward_raw(x, w, tabs[0][1], n)
        torch.cuda.synchronize()
        res = {0: [], 1: [], 2: [], 3: [], 4: []}
        for rep in range(10):
            for arm in (0, 1, 2, 3, 4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(40):
                    ys[arm] = sparse._conv_forward_raw(x, w, tabs[arm][1], n)
                e1.record()
                torch.cuda.synchronize()
                res[arm].append(e0.elapsed_time(e1) / 40 * 1e3)
        print('L%d %dx%d N=%d: numeric order %.1f us (tile fill %.3f) | geometric rarest-first %.1f us (%.3f) | per-chunk rarest-first '
              '%.1f us (%.3f) | 8192-row chunks %.1f us (%.3f) | 16384-row chunks %.1f us (%.3f) | results equal: %s' % (
                  lvl, c, c, n, np.median(res[0]), tile_fill(tabs[0][1], n), np.median(res[1]), tile_fill(tabs[1][1], n),
                  np.median(res[2]), tile_fill(tabs[2][1], n), np.median(res[3]), tile_fill(tabs[3][1], n),
                  np.median(res[4]), tile_fill(tabs[4][1], n), all(bool(torch.equal(ys[0], ys[a])) for a in (1, 2, 3, 4))),
              flush=True)
