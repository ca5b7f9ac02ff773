#!/usr/bin/env python
"""Micro-benchmark of the sparse-conv kernels on the real SECOND bs=16 geometry (levels 1..4 of VoxelBackBone8x).
Prints per (level, Cin, Cout): rows, pairs, time, algorithmic GB/s (SURVEY §8d bytes), f32 TFLOP/s for fwd and wgrad.
Usage: python tools/bench_sparse_conv.py [--batch 16] [--iters 50] [--no-sort]"""
import argparse
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def balg_of(n, cin, cout, P):
    return 4.0 * n * cin + 4.0 * n * cout + 8.0 * P + 4.0 * 27 * cin * cout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--no-sort', action='store_true')
    ap.add_argument('--no-lpt', action='store_true')
    ap.add_argument('--levels', default='1,2,3,4')
    ap.add_argument('--chunk', type=int, default=None)
    ap.add_argument('--dense-random', action='store_true', help='all 27 neighbours present, rows drawn at random inside +-4096 rows (real-table-like locality)')
    ap.add_argument('--dense-k', type=int, default=27, help='with --dense: only the first k offsets are present in every row')
    ap.add_argument('--no-windowed', action='store_true', help='wgrad: per-offset pair ranges instead of output-row windows')
    ap.add_argument('--wgrad-sweep', action='store_true', help='sweep the workgroups-per-offset knob of the wgrad')
    ap.add_argument('--dense', action='store_true', help='synthetic table with all 27 neighbours present (no skip imbalance)')
    args = ap.parse_args()
    from crbhip import sparse, voxel
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    if args.no_windowed:
        sparse.WGRAD_WINDOWED = False
    if args.no_sort:
        sparse.MASK_SORT = False
    if args.no_lpt:
        sparse.TILE_LPT = False
    if args.chunk is not None:
        sparse.MASK_SORT_CHUNK = args.chunk
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, args.batch)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    chans = {1: (16, 16), 2: (32, 32), 3: (64, 64), 4: (64, 64)}
    levels = [int(v) for v in args.levels.split(',')]
    for lvl in range(1, 5):
        if lvl > 1:
            ks, st, pd = geo[lvl - 2]
            rbs = sparse.spconv_rulebook(coords, shape, args.batch, ks, st, pd)
            coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        if lvl not in levels:
            continue
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        cin, cout = chans[lvl]
        n = rb.n_out
        P = int((rb.nbr >= 0).sum())
        x = torch.randn(n, cin, device=dev)
        dy = torch.randn(n, cout, device=dev)
        w = torch.randn(27, cin, cout, device=dev) / 10
        table = rb.table_for('nbr', cin, cout) if not os.environ.get('CRB_NO_COMPACT') else rb.sorted_table('nbr')
        pairs = rb.pairs()
        if args.dense_random:
            g = torch.Generator(device=dev).manual_seed(0)
            ar = torch.arange(n, device=dev, dtype=torch.int64).view(-1, 1)
            jit = torch.randint(-4096, 4096, (n, 27), device=dev, generator=g)
            dense_nbr = ((ar + jit) % n).to(torch.int32).contiguous()
            if args.dense_k < 27:
                dense_nbr[:, args.dense_k:] = -1
            table = (dense_nbr, None)
            P = min(27, args.dense_k) * n
        if args.dense:
            ar = torch.arange(n, device=dev, dtype=torch.int64).view(-1, 1)
            dense_nbr = ((ar + torch.arange(27, device=dev, dtype=torch.int64).view(1, -1) * 97) % n).to(torch.int32).contiguous()
            if args.dense_k < 27:
                dense_nbr[:, args.dense_k:] = -1
            table = (dense_nbr, None)
            P = min(27, args.dense_k) * n

        def timeit(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3      # us

        from crbhip import lib
        ts = {}
        for st_ in (1, 16, 8):                                     # v1 (1 / 2 row tiles per wave), v2
            lib.crb_sparse_conv_set_subtiles(st_)
            ts[st_] = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
        lib.crb_sparse_conv_set_subtiles(0)
        t_f = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
        t_fk = t_f
        if isinstance(table, sparse.CompactTable):
            tk_ = rb.sorted_table('nbr')
            t_fk = timeit(lambda: sparse._conv_forward_raw(x, w, tk_, n))
            assert torch.equal(sparse._conv_forward_raw(x, w, tk_, n), sparse._conv_forward_raw(x, w, table, n))
        bx = ''
        if isinstance(table, sparse.CompactTable) and lib.crb_sparse_conv_bf16x3_supported(cin, cout):
            y3 = sparse._conv_forward_raw(x, w, table, n, arithmetic='bf16x3')
            tb = []
            for tpw in (1, 2, 3):
                lib.crb_sparse_conv_bf16x3_set_tiles_per_wave(tpw)
                tb.append(timeit(lambda: sparse._conv_forward_raw(x, w, table, n, arithmetic='bf16x3')))
            if cin == 64 and cout == 64:
                for tpw in (1, 2):
                    lib.crb_sparse_conv_bf16x3_set_tiles_per_wave(tpw)
                    tm = []
                    for mode in (1, 2, 3, 4, 5, 0):
                        lib.crb_sparse_conv_bf16x3_set_mode(mode)
                        tm.append(timeit(lambda: sparse._conv_forward_raw(x, w, table, n, arithmetic='bf16x3')))
                    lib.crb_sparse_conv_bf16x3_set_mode(0)
                    print('   bf16x3 tpw%d measurement builds: no-MFMA %.1f, no-gather %.1f, no-W %.1f, no-gather-no-W %.1f, nt gathers %.1f, normal %.1f us' % (
                        tpw, *tm))
            lib.crb_sparse_conv_bf16x3_set_tiles_per_wave(0)
            yf = sparse._conv_forward_raw(x, w, table, n)
            S = sparse._conv_forward_raw(x.abs(), w.abs(), table, n)
            rel = float(((y3 - yf).abs() / S.clamp_min(1e-30)).max())
            bx = ' | bf16x3 fwd (incl. W split) tpw1 %.1f us, tpw2 %.1f us, 8 waves x 1 tile %.1f us = %.0f GB/s alg (%.1f%% of 8TB/s), max |d|/sum|x||w| = 2^%.1f' % (
                tb[0], tb[1], tb[2], balg_of(n, cin, cout, P) / min(tb) / 1e3, balg_of(n, cin, cout, P) / min(tb) / 1e3 / 80,
                np.log2(max(rel, 1e-30)))
        t_w = timeit(lambda: sparse._conv_wgrad_raw(x, dy, pairs, 27))
        lib.crb_sparse_conv_set_wgrad_v1(1)
        t_w1 = timeit(lambda: sparse._conv_wgrad_raw(x, dy, pairs, 27))
        dw1 = sparse._conv_wgrad_raw(x, dy, pairs, 27)
        lib.crb_sparse_conv_set_wgrad_v1(0)
        dw2 = sparse._conv_wgrad_raw(x, dy, pairs, 27)
        werr = float((dw2 - dw1).abs().max() / dw1.abs().max())
        sweep = ''
        if args.wgrad_sweep:
            for sp_ in (16, 24, 40, 64, 96, 160):
                lib.crb_sparse_conv_set_wgrad_splits(sp_)
                sweep += ' S%d=%.1f' % (sp_, timeit(lambda: sparse._conv_wgrad_raw(x, dy, pairs, 27)))
            lib.crb_sparse_conv_set_wgrad_splits(0)
        balg = 4.0 * n * cin + 4.0 * n * cout + 8.0 * P + 4.0 * 27 * cin * cout
        fl = 2.0 * P * cin * cout
        print('L%d subm %dx%d N=%d P=%d (%.2f/row) | fwd %.1f us  %.0f GB/s alg (%.1f%% of 8TB/s)  %.1f TF | '
              'wgrad %.1f us %.1f TF (%.1f%% of 157.3) [v1 kernel %.1f us, max rel diff %.1e]%s' % (
                  lvl, cin, cout, n, P, P / n, t_f, balg / t_f / 1e3, balg / t_f / 1e3 / 80, fl / t_f / 1e6, t_w,
                  fl / t_w / 1e6, fl / t_w / 1e6 / 1.573, t_w1, werr, sweep),
              'v1/v2-noremap/v2 %.1f %.1f %.1f us | (n,K) table fwd %.1f us' % (ts[1], ts[16], ts[8], t_fk) + bx, flush=True)


if __name__ == '__main__':
    main()
