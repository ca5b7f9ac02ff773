#!/usr/bin/env python
"""where the low-channel subm gather-GEMMs (4->16, 16->16 at level 1; 32->32 at level 2; wgrad 32x32) stand
against what a launch of their size CAN reach on this chip. For every kernel: median time (300 warm-up launches, 12 x 40
launches), algorithmic bytes, and three floors measured in the same process with the same protocol:
  copy      a plain f32 copy kernel (torch) moving the same algorithmic bytes: the HBM/launch floor of a transfer this small
  gather    torch.index_select of P rows of C_in floats + the write of N x C_out (what a gather of the pair list costs without
            any arithmetic, weights or table decoding)
  empty     an empty-ish launch (fill of 256 floats): the launch + dispatch floor
usage: python tools/lowchannel_floor.py"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # A/B knob of the low-channel kernel (include/crb_hip_measure.h)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timeit(fn, warm=300, rounds=12, per=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(per):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / per)
    return float(np.median(ts))


if __name__ == '__main__':
    from crbhip import sparse, voxel, lib
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    t_empty = timeit(lambda: torch.empty(256, device=dev).fill_(0.0))
    print('empty launch (fill of 256 floats): %.2f us per launch back to back' % t_empty)
    geo = ((3, 3, 3), (2, 2, 2), (1, 1, 1))
    for level, pairs_c in ((1, ((4, 16), (16, 16))), (2, ((32, 32),))):
        if level == 2:
            rbs = sparse.spconv_rulebook(coords, shape, 16, *geo)
            coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        n = rb.n_out
        P = int((rb.nbr >= 0).sum())
        pin = rb.nbr[rb.nbr >= 0].long()
        for cin, cout in pairs_c:
            x = torch.randn(n, cin, device=dev)
            w = torch.randn(27, cin, cout, device=dev) / 10
            dy = torch.randn(n, cout, device=dev)
            table = rb.table_for('nbr', cin, cout, 'f32')
            balg = 4.0 * n * cin + 4.0 * n * cout + 8.0 * P + 4.0 * 27 * cin * cout
            flops = 2.0 * P * cin * cout
            t_k = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
            src = torch.empty(int(balg // 8), device=dev)
            dst = torch.empty_like(src)
            t_copy = timeit(lambda: dst.copy_(src))
            yout = torch.empty(n, cout, device=dev)

            def gather():
                torch.index_select(x, 0, pin)
                yout.fill_(0.0)
            t_g = timeit(gather)
            line = ('level %d subm %2d->%2d: N = %d, P = %d (%.1f per row), algorithmic %.1f MB, %.2f GF | kernel %.1f us = %.0f GB/s = %.1f %% of 8 TB/s, '
                    '%.1f TF = %.1f %% of the f32 MFMA roof | floors: copy of the same bytes %.1f us (%.0f GB/s), gather P rows + write N rows '
                    '(2 launches) %.1f us, empty launch %.1f us'
                    % (level, cin, cout, n, P, P / n, balg / 1e6, flops / 1e9, t_k, balg / t_k / 1e3, 100 * balg / t_k / 1e3 / 8000,
                       flops / t_k / 1e6, 100 * flops / t_k / 1e6 / 157.3, t_copy, balg / t_copy / 1e3, t_g, t_empty))
            print(line, flush=True)
            if cout == 16:
                # round 4: resident-weights kernel (sparse_conv_fwd_lc_kernel) against the phase kernels it replaces, same inputs
                lib.crb_sparse_conv_set_lowchannel(2)
                y_new = sparse._conv_forward_raw(x, w, table, n)
                lib.crb_sparse_conv_set_lowchannel(0)
                old_table = table if cin == 16 else rb.sorted_table('nbr')
                y_old = sparse._conv_forward_raw(x, w, old_table, n)
                t_old = timeit(lambda: sparse._conv_forward_raw(x, w, old_table, n))
                grids = {}
                for g in (2, 256, 512):
                    lib.crb_sparse_conv_set_lowchannel(g)
                    grids[g] = timeit(lambda: sparse._conv_forward_raw(x, w, table, n), warm=100, rounds=8, per=40)
                lib.crb_sparse_conv_set_lowchannel(1)
                print('   product path %.1f us | phase / v1 kernel %.1f us; low-channel kernel bit-equal to it: %s; weights from L2 + one tile per wave / grid cap 256 / 512 workgroups: %s us'
                      % (t_k, t_old, bool(torch.equal(y_new, y_old)), ' / '.join('%.1f' % grids[g] for g in grids)), flush=True)
            if cin >= 32:
                prs = rb.pairs()
                t_w = timeit(lambda: sparse._conv_wgrad_raw(x, dy, prs, 27), warm=100, rounds=8, per=20)
                print('level %d wgrad %2dx%2d: %.1f us = %.1f TF = %.1f %% of the f32 MFMA roof, %.0f GB/s algorithmic'
                      % (level, cin, cout, t_w, flops / t_w / 1e6, 100 * flops / t_w / 1e6 / 157.3, balg / t_w / 1e3), flush=True)
