#!/usr/bin/env python
"""VERDICT r05 item 5, third floor file: the INPUT-STATIONARY formulation of the level-1 subm layers (16-channel rows, 3.5 neighbours per
row) as a probe - every input row read once, added into the output rows of its neighbours (csrc/probe_floor.hip variants 3 / 4: global
float atomics; the same with the workgroup's own 64 output rows collected in LDS first). No weights, no MFMA: the memory side only, as
in tools/lowchannel_floor2.py, same table, same protocol (300 warm-up launches, 12 x 40 launches, median). The output has to be zero
before the launch: timed with and without the clearing pass. usage: python tools/lowchannel_floor3.py"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
from lowchannel_floor import timeit  # noqa: E402

if __name__ == '__main__':
    from crbhip import sparse, voxel, lib, check, ptr, cur_stream
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
    n = rb.n_out
    table = rb.table_for('nbr', 16, 16, 'f32')
    nbr = table.to_nbr()
    P = int((nbr >= 0).sum())
    order = torch.argsort((nbr < 0).int(), dim=1, stable=True)
    ell = torch.gather(nbr, 1, order)[:, :8].contiguous().int()
    x = torch.randn(n, 16, device=dev)
    y = torch.empty(n, 16, device=dev)
    w = torch.randn(27, 16, 16, device=dev) / 10
    st = cur_stream(dev)
    # correctness of the probes: y = sum over the rows that list i ... = scatter of x along ell
    want = torch.zeros(n, 16, device=dev)
    src = torch.arange(n, device=dev)[:, None].expand(n, 8)[ell >= 0]
    want.index_add_(0, ell[ell >= 0].long(), x[src])
    for v in (3, 4):
        y.zero_()
        check(lib.crb_probe_gather_chain(v, ptr(x), n, None, None, None, ptr(ell), ptr(y), st), 'probe')
        assert torch.allclose(y, want, atol=1e-4), v
    inside = int(((ell >= 0) & ((ell // 64) == (torch.arange(n, device=dev)[:, None] // 64))).sum())
    t_empty = timeit(lambda: torch.empty(256, device=dev).fill_(0.0))
    t_zero = timeit(lambda: y.zero_())
    res = {}
    for v in (0, 1, 3, 4):
        res[v] = timeit(lambda v=v: check(lib.crb_probe_gather_chain(v, ptr(x), n, None, None, None, ptr(ell), ptr(y), st), 'probe'))
    res['3z'] = timeit(lambda: (y.zero_(), check(lib.crb_probe_gather_chain(3, ptr(x), n, None, None, None, ptr(ell), ptr(y), st), 'probe')))
    res['4z'] = timeit(lambda: (y.zero_(), check(lib.crb_probe_gather_chain(4, ptr(x), n, None, None, None, ptr(ell), ptr(y), st), 'probe')))
    t_k = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
    balg = 4.0 * n * 16 * 2 + 8.0 * P + 4.0 * 27 * 256
    pairs8 = int((ell >= 0).sum())
    print('level-1 subm table: N = %d rows, P = %d pairs (%.2f per row), %d in the fixed-stride lists; %.1f %% of the listed neighbours lie in the '
          'lister\'s own 64-row block' % (n, P, P / n, pairs8, 100.0 * inside / pairs8))
    print('empty launch %.2f us, clearing the output (%.1f MB) %.2f us' % (t_empty, 64.0 * n / 1e6, t_zero))
    print('probe 0: copy (1 round trip)                                                              %6.2f us' % res[0])
    print('probe 1: gather - fixed-stride list -> rows -> store (2 dependent round trips)            %6.2f us' % res[1])
    print('probe 3: input-stationary - row + list once -> float atomics into the neighbours\' rows     %6.2f us (%.2f with the clearing pass)'
          % (res[3], res['3z']))
    print('probe 4: the same, own 64-row block collected in LDS first, the rest by global atomics    %6.2f us (%.2f with the clearing pass)'
          % (res[4], res['4z']))
    print('product kernel on this table (sparse_conv_fwd2_kernel<16,16>): %.2f us = %.1f %% of 8 TB/s on %.1f MB algorithmic' %
          (t_k, 100 * balg / t_k / 1e3 / 8000, balg / 1e6))
    print('the target of the review (40 %% of 8 TB/s) = %.2f us' % (balg / 0.4 / 8e6))
