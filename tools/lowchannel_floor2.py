#!/usr/bin/env python
"""VERDICT r04 item 4, second floor file: what the DEPENDENT-round-trip chain of a gather costs at the size of the level-1 subm
layers (16-channel rows, 3.5 neighbours per row), measured with probe kernels that do the memory side of the chain and nothing else
(csrc/probe_floor.hip, measurement library): copy (1 round trip), fixed-stride neighbour list -> rows (2), compact table -> packed
indices -> rows (3). Same protocol as tools/lowchannel_floor.py (300 warm-up launches, 12 x 40 launches, median); the product
kernels on the same table beside them. usage: python tools/lowchannel_floor2.py"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np  # noqa: E402
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
    assert isinstance(table, sparse.CompactTable)
    nbr = table.to_nbr()                                              # (n, 27) in kernel order
    P = int((nbr >= 0).sum())
    cnt = (nbr >= 0).sum(1)
    # fixed-stride list: the first 8 present neighbours of every row (rows with more keep the first 8: 16-channel rows only need
    # the chain, the probe is about latency, not about the sum)
    order = torch.argsort((nbr < 0).int(), dim=1, stable=True)
    ell = torch.gather(nbr, 1, order)[:, :8].contiguous().int()
    x = torch.randn(n, 16, device=dev)
    y = torch.empty(n, 16, device=dev)
    w = torch.randn(27, 16, 16, device=dev) / 10
    st = cur_stream(dev)
    print('level-1 subm table: N = %d rows, P = %d pairs (%.2f per row; %.1f %% of the rows have more than 8), rows of 16 floats'
          % (n, P, P / n, 100.0 * float((cnt > 8).float().mean())))
    t_empty = timeit(lambda: torch.empty(256, device=dev).fill_(0.0))
    res = {}
    for v, name in ((0, 'copy (1 round trip)'), (1, 'fixed-stride list -> rows (2 round trips)'),
                    (2, 'cmask/cbase -> packed -> rows (3 round trips)')):
        res[v] = timeit(lambda v=v: check(lib.crb_probe_gather_chain(v, ptr(x), n, ptr(table.cmask), ptr(table.cbase), ptr(table.packed),
                                                                    ptr(ell), ptr(y), st), 'probe'))
    ref = torch.zeros_like(y)
    check(lib.crb_probe_gather_chain(1, ptr(x), n, None, None, None, ptr(ell), ptr(y), st), 'probe')
    g = torch.where(ell[:, :, None] >= 0, x[ell.clamp(min=0).long()], torch.zeros((), device=dev)).sum(1)
    assert torch.allclose(y, g, atol=1e-5)
    t_k = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
    lib.crb_sparse_conv_set_lowchannel(2)
    t_lc = timeit(lambda: sparse._conv_forward_raw(x, w, table, n))
    lib.crb_sparse_conv_set_lowchannel(1)
    balg = 4.0 * n * 16 * 2 + 8.0 * P + 4.0 * 27 * 256
    print('empty launch %.2f us' % t_empty)
    for v, name in ((0, 'copy (1 round trip)'), (1, 'fixed-stride neighbour list -> rows -> store (2 dependent round trips)'),
                    (2, 'cmask / cbase -> packed indices -> rows -> store (3 dependent round trips)')):
        print('probe %d: %-75s %6.2f us' % (v, name, res[v]))
    print('product kernels on this table: phase kernel (sparse_conv_fwd2_kernel<16,16>) %.2f us = %.1f %% of 8 TB/s on %.1f MB algorithmic; '
          'wave-owned tiles (sparse_conv_fwd_lc_kernel, 3 round trips) %.2f us' % (t_k, 100 * balg / t_k / 1e3 / 8000, balg / 1e6, t_lc))
    gath = 64.0 * min(P, int(cnt.clamp(max=8).sum())) + 64.0 * n + 32.0 * n
    print('bytes the two-round-trip probe moves through the cache hierarchy: %.1f MB (%.1f MB of gathered rows: every input row is fetched '
          '%.2f times, + the output + the list) in %.2f us beyond the launch floor = %.1f TB/s' % (
              gath / 1e6, 64.0 * P / 1e6, P / n, res[1] - t_empty, gath / (res[1] - t_empty) / 1e6))
    print('the target of VERDICT r04 item 4 (40 %% of 8 TB/s) = %.2f us; 60 %% (north_star) = %.2f us' % (balg / 0.4 / 8e6, balg / 0.6 / 8e6))
