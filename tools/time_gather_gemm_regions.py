#!/usr/bin/env python
"""Where the 64x64 gather-GEMM spends its cycles: s_memtime accounting per wave (measurement build of the kernel,
crb_sparse_conv_set_subtiles(32)) on the SECOND bs=16 level-3/4 tables and on a table with all 27 neighbours present."""
import ctypes
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

NAMES = ['waves', 'total', 'prologue 3: first W', 'gather issue', 'MFMA block', 'W store(+vmcnt)', 'barrier wait', 'epilogue', 'phases',
         'phases with MFMA', 'W fetch issue', 'row index (LDS)', 'prefetch landed?', 'prologue 1: table', 'prologue 2: masks']


def main():
    from crbhip import sparse, voxel, lib
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    for lvl in range(2, 5):
        rbs = sparse.spconv_rulebook(coords, shape, 16, *geo[lvl - 2])
        coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        if lvl < 3:
            continue
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        n = rb.n_out
        x = torch.randn(n, 64, device=dev)
        w = torch.randn(27, 64, 64, device=dev) / 10
        tables = {'real L%d' % lvl: rb.sorted_table('nbr')}
        if lvl == 3:
            ar = torch.arange(n, device=dev, dtype=torch.int64).view(-1, 1)
            tables['all 27 present'] = (((ar + torch.arange(27, device=dev).view(1, -1) * 97) % n).to(torch.int32).contiguous(), None)
        for name, table in tables.items():
            lib.crb_sparse_conv_set_subtiles(32)
            buf = (ctypes.c_uint64 * 16)()
            lib.crb_sparse_conv_timing(buf)                     # clear
            for _ in range(3):
                sparse._conv_forward_raw(x, w, table, n)
            torch.cuda.synchronize()
            lib.crb_sparse_conv_timing(buf)
            lib.crb_sparse_conv_set_subtiles(0)
            v = [int(b) for b in buf]
            waves = max(v[0], 1)
            print('== %s: %d waves, %.1f phases/wave (%.1f with MFMA work), %.0f cycles/wave (s_memtime ticks = shader cycles)'
                  % (name, waves // 3, v[8] / waves, v[9] / waves, v[1] / waves))
            for k in (13, 14, 2, 12, 10, 11, 3, 4, 5, 6, 7):
                print('   %-18s %8.0f cycles/wave  %5.1f %%   %7.0f per phase' % (NAMES[k], v[k] / waves, 100.0 * v[k] / v[1],
                                                                                v[k] / max(v[8], 1)))


if __name__ == '__main__':
    main()
