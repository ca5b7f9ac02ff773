"""GPU: every entry point of csrc/sa_mlp_train.hip (training-mode set abstraction as recompute passes) against a
float64 torch restatement of the module lines it replaces (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py:90-108 with
train-mode BatchNorm: group -> Conv 1x1 -> BN -> ReLU -> Conv 1x1 -> BN -> ReLU -> max over nsample) and of their autograd, one entry
point at a time, on edge cases of the work distribution: query counts that are not multiples of the 16-query chunks, frames with no
live query, no empty ball at all, every ball empty, nsample 16 / 32 / 48, all width pairs the RoI-grid / VSA layers use.
Tolerance: 2e-5 of the largest reference entry (observed <= 2e-6: f32 sums in another order)."""
import ctypes

import numpy as np
import pytest
import torch

from synth import kitti_batch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


CASES = [
    # h1, h2, ns, queries per frame, which queries are pushed out of range (empty balls): 'third' / 'none' / 'all' / 'frame1'
    (64, 64, 16, (500, 301), 'third'),
    (32, 16, 32, (37, 5), 'third'),
    (16, 64, 16, (1, 1), 'none'),
    (64, 32, 48, (130, 77), 'frame1'),
    (16, 16, 16, (640, 0), 'none'),
    (32, 32, 16, (90, 90), 'all'),
    (64, 64, 32, (2049, 17), 'third'),
]


@pytest.mark.parametrize('h1,h2,ns,counts,empties', CASES)
def test_entry_points_against_the_float64_restatement(dev, h1, h2, ns, counts, empties):
    from crbhip import lib, check, ptr, cur_stream, bnrelu
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    assert lib.crb_sa_mlp2_train_supported(h1, h2, ns)
    # every ball empty: all rows equal, the batch variance is 0 and invstd = 1 / sqrt(eps) = 316 multiplies the f32 rounding of
    # (y - mean): a degenerate BatchNorm, checked at 100 x the tolerance
    TOL = 2e-3 if empties == 'all' else globals()['TOL']
    torch.manual_seed(h1 + h2 + ns)
    pts, off, _ = kitti_batch(2, 7, n_points=6000)
    xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).to(dev)
    xc = torch.from_numpy(np.diff(off).astype(np.int32)).to(dev)
    rng = np.random.default_rng(2)
    sel = np.concatenate([k * 6000 + rng.choice(6000, n, replace=n > 6000) for k, n in enumerate(counts)]).astype(np.int64)
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    if empties == 'third':
        new[::3] += 55.0
    elif empties == 'all':
        new += 55.0
    elif empties == 'frame1':
        new[counts[0]:] += 55.0
    nc = torch.tensor(counts, dtype=torch.int32, device=dev)
    C = 20
    feat = torch.randn(12000, C, device=dev)
    W1 = torch.randn(h1, 3 + C, device=dev) * 0.3
    W2 = torch.randn(h2, h1, device=dev) * 0.3
    g1, b1 = torch.randn(h1, device=dev) * 0.5 + 0.8, torch.randn(h1, device=dev) * 0.3
    g2, b2 = torch.randn(h2, device=dev) * 0.5 + 0.8, torch.randn(h2, device=dev) * 0.3
    idx, empty = U.ball_query(1.2, ns, xyz, xc, new, nc)
    M, B = new.shape[0], 2
    n = M * ns
    em = empty.bool()
    assert {'none': not bool(em.any()), 'all': bool(em.all()), 'third': bool(em.any()) and not bool(em.all()),
            'frame1': bool(em[counts[0]:].all())}[empties]
    st = cur_stream(dev)
    # ---- float64 restatement
    start = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), xc.long().cumsum(0)[:-1]])
    qframe = torch.repeat_interleave(torch.arange(B, device=dev), nc.long())
    rows = start[qframe][:, None] + idx.long()
    grp = torch.cat([(xyz[rows] - new[:, None, :]).double(), feat[rows].double()], 2)
    grp[em] = 0
    x = grp.reshape(n, 3 + C)
    y1 = x @ W1.double().t()
    mu1, is1 = y1.mean(0), (y1.var(0, unbiased=False) + 1e-5).rsqrt()
    xh1 = (y1 - mu1) * is1
    z1 = (g1.double() * xh1 + b1.double()).clamp_min(0)
    y2 = z1 @ W2.double().t()
    mu2, is2 = y2.mean(0), (y2.var(0, unbiased=False) + 1e-5).rsqrt()
    xh2 = (y2 - mu2) * is2
    z2 = (g2.double() * xh2 + b2.double()).clamp_min(0)
    out_ref = z2.view(M, ns, h2).max(1).values
    # ---- forward entry points
    w1x, w1f = W1[:, :3].t().contiguous(), W1[:, 3:].contiguous()
    P = feat @ w1f.t()
    emu = empty.to(torch.uint8).contiguous()
    nslab = int(lib.crb_group_affine_rows_grad_blocks(M, ns))
    slab1 = torch.empty((nslab, 2, h1), device=dev)
    check(lib.crb_group_affine_rows_stats_stack(B, M, h1, ns, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu), ptr(w1x),
                                                None, None, ptr(slab1), st), 'pass 0')
    s = slab1.double().sum(0)
    assert _rel(s[0] / n, mu1) <= TOL and _rel(s[1] / n, (y1 * y1).mean(0)) <= TOL
    mean1, invstd1 = mu1.float().contiguous(), is1.float().contiguous()
    nwave = int(lib.crb_sa_mlp2_train_waves(M))
    slab2 = torch.full((nwave, 2, h2), float('nan'), device=dev)
    check(lib.crb_sa_mlp2_train_stats(B, M, ns, h1, h2, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu), ptr(w1x),
                                      ptr(mean1), ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(slab2), st), 'pass A')
    s = slab2.double().sum(0)                                  # (every wave of the launch wrote its partial: no NaN left)
    assert _rel(s[0] / n, mu2) <= TOL and _rel(s[1] / n, (y2 * y2).mean(0)) <= TOL
    mean2, invstd2 = mu2.float().contiguous(), is2.float().contiguous()
    out = torch.full((M, h2 + 8), float('nan'), device=dev)
    arg = torch.full((M, h2), -7, dtype=torch.int32, device=dev)
    ysel = torch.full((M, h2), float('nan'), device=dev)
    check(lib.crb_sa_mlp2_train_max(B, M, ns, h1, h2, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu), ptr(w1x), ptr(mean1),
                                    ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2),
                                    ctypes.c_void_p(out.data_ptr() + 16), h2 + 8, ptr(arg), ptr(ysel), st), 'pass B')
    o = out[:, 4:4 + h2]
    assert bool(torch.isnan(out[:, :4]).all()) and bool(torch.isnan(out[:, 4 + h2:]).all())        # only its own columns are written
    assert int(arg.min()) >= 0 and int(arg.max()) < ns
    z_at_arg = z2.view(M, ns, h2).gather(1, arg.long()[:, None, :])[:, 0]
    y_at_arg = y2.view(M, ns, h2).gather(1, arg.long()[:, None, :])[:, 0]
    assert _rel(o, out_ref) <= TOL
    assert float((z_at_arg - out_ref).abs().max()) <= 1e-6 * float(out_ref.abs().max())            # the arg attains the maximum
    assert _rel(ysel, y_at_arg) <= TOL
    assert int(arg[em].abs().sum()) == 0                                                            # empty balls: equal rows, first wins
    # ---- backward
    go = torch.randn(M, h2 + 8, device=dev)
    dz2 = torch.zeros(M, ns, h2, dtype=torch.float64, device=dev)
    dz2.scatter_(1, arg.long()[:, None, :], (go[:, 4:4 + h2].double() * (z_at_arg > 0))[:, None, :])
    dz2 = dz2.view(n, h2)
    db2, dg2 = dz2.sum(0), (dz2 * xh2).sum(0)
    dy2 = g2.double() * is2 * (dz2 - db2 / n - xh2 * dg2 / n)
    dz1 = (dy2 @ W2.double()) * (z1 > 0)
    dW2_ref = dy2.t() @ z1
    db1, dg1 = dz1.sum(0), (dz1 * xh1).sum(0)
    dy1 = g1.double() * is1 * (dz1 - db1 / n - xh1 * dg1 / n)
    dW1_ref = dy1.t() @ x
    gfeat_ref = torch.zeros(12000, C, dtype=torch.float64, device=dev)
    live_rows = (~em)[:, None].expand(M, ns).reshape(-1)
    gfeat_ref.index_add_(0, rows.reshape(-1)[live_rows], (dy1 @ W1.double())[live_rows][:, 3:])
    d2 = torch.empty((2, h2), device=dev)
    wsb = lib.crb_bn_workspace_bytes(M, h2)
    ws, tk = bnrelu._scratch(dev, wsb)
    gp = ctypes.c_void_p(go.data_ptr() + 16)
    bnrelu._bn_check(lib.crb_bn_relu_max_backward_sums(ptr(ysel), gp, h2 + 8, M, h2, ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2), ptr(d2[1]),
                                                       ptr(d2[0]), ptr(ws), wsb, ptr(tk), st), 'sums')
    scale2 = max(float(dz2.abs().sum(0).max()), float((dz2 * xh2).abs().sum(0).max()), 1e-30)
    assert float((d2[0].double() - db2).abs().max()) <= TOL * scale2 and float((d2[1].double() - dg2).abs().max()) <= TOL * scale2
    gz1 = torch.full((n, h1), float('nan'), device=dev)
    d1 = torch.empty((2, h1), device=dev)
    dW2 = torch.empty((h2, h1), device=dev)
    wsf = int(lib.crb_sa_mlp2_train_backward_workspace_floats(M, h1, h2))
    wsp = torch.empty((wsf,), device=dev)
    db2f, dg2f = db2.float().contiguous(), dg2.float().contiguous()
    check(lib.crb_sa_mlp2_train_backward(B, M, ns, h1, h2, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu), ptr(w1x), ptr(mean1),
                                         ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2), gp, h2 + 8,
                                         ptr(arg), ptr(db2f), ptr(dg2f), ptr(gz1), ptr(d1), ptr(dW2), ptr(wsp), wsf, st), 'pass C')
    gz1v = gz1.view(M, ns, h1)
    if bool((~em).any()):
        assert _rel(gz1v[~em], dz1.view(M, ns, h1)[~em]) <= TOL
    assert bool(torch.isnan(gz1v[em]).all())                     # rows of empty balls are not written (their sums are folded in)
    # BatchNorm-1 sums: sums of signed terms that cancel (exactly, when every row is equal): measured against the sum of magnitudes
    s1 = max(float(dz1.abs().sum(0).max()), float((dz1 * xh1).abs().sum(0).max()), 1e-30)
    assert float((d1[0].double() - db1).abs().max()) <= TOL * s1 and float((d1[1].double() - dg1).abs().max()) <= TOL * s1
    sW = float((dy2.abs().t() @ z1.abs()).max())
    assert float((dW2.double() - dW2_ref).abs().max()) <= TOL * max(sW, 1e-30)
    gP = torch.zeros((12000, h1), device=dev)
    part = torch.empty((nslab, 3, h1), device=dev)
    db1f, dg1f = db1.float().contiguous(), dg1.float().contiguous()
    check(lib.crb_group_affine_rows_grad_bn_recompute_stack(B, M, h1, ns, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu),
                                                            ptr(w1x), ptr(gz1), ptr(mean1), ptr(invstd1), ptr(g1), ptr(b1), ptr(db1f),
                                                            ptr(dg1f), None, None, 0, ptr(gP), ptr(part), st), 'pass D')
    assert bool(torch.isfinite(part).all()) and bool(torch.isfinite(gP).all())
    gW1 = torch.cat([part.sum(0).t(), gP.t() @ feat], 1)
    if bool((~em).any()):
        assert _rel(gW1, dW1_ref) <= 10 * TOL and _rel(gP @ w1f, gfeat_ref) <= 10 * TOL
    else:
        assert float(gW1.abs().max()) == 0.0 and float(gP.abs().max()) == 0.0
    # the same scatter in SOURCE-ROW order (crb_pair_sort_by_source): a stable sort of the pairs, pairs of empty balls last
    sp = torch.full((n,), -1, dtype=torch.int32, device=dev)
    sr = torch.full((n,), -1, dtype=torch.int32, device=dev)
    wsb = int(lib.crb_pair_sort_workspace_bytes(M, ns))
    wss = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    check(lib.crb_pair_sort_by_source(B, M, ns, ptr(xc), ptr(nc), ptr(idx), ptr(emu), 12000, ptr(sp), ptr(sr), ptr(wss), wsb, st), 'sort')
    key = torch.where(em[:, None].expand(M, ns), torch.full_like(rows, 12000), rows).reshape(-1)
    order = torch.sort(key, stable=True)[1]
    assert torch.equal(sp.long(), order) and torch.equal(sr.long(), key[order])
    gP2 = torch.zeros((12000, h1), device=dev)
    part2 = torch.full((nslab, 3, h1), float('nan'), device=dev)
    check(lib.crb_group_affine_rows_grad_bn_recompute_stack(B, M, h1, ns, ptr(xyz), ptr(xc), ptr(P), ptr(new), ptr(nc), ptr(idx), ptr(emu),
                                                            ptr(w1x), ptr(gz1), ptr(mean1), ptr(invstd1), ptr(g1), ptr(b1), ptr(db1f),
                                                            ptr(dg1f), ptr(sp), ptr(sr), 12000, ptr(gP2), ptr(part2), st), 'pass D, sorted')
    assert bool(torch.isfinite(part2).all())
    gW1s = torch.cat([part2.sum(0).t(), gP2.t() @ feat], 1)
    if bool((~em).any()):
        assert _rel(gW1s, dW1_ref) <= 10 * TOL and _rel(gP2 @ w1f, gfeat_ref) <= 10 * TOL
        assert float((gP2 - gP).abs().max()) <= 1e-5 * max(1.0, float(gP.abs().max()))
    else:
        assert float(gW1s.abs().max()) == 0.0 and float(gP2.abs().max()) == 0.0


def test_unsupported_shapes_say_so_and_the_module_keeps_the_rows_path(dev):
    from crbhip import lib
    from pcdet.config import EasyDict
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    assert not lib.crb_sa_mlp2_train_supported(128, 64, 16) and not lib.crb_sa_mlp2_train_supported(64, 64, 8)
    assert not lib.crb_sa_mlp2_train_supported(64, 48, 16) and lib.crb_sa_mlp2_train_supported(16, 64, 48)
    layer, _ = M.build_local_aggregation_module(20, EasyDict({'MLPS': [[32, 32, 32]], 'POOL_RADIUS': [0.8], 'NSAMPLE': [16]}))
    assert not layer.to(dev).train()._train_fused_ok()           # three layers: rows path
    layer, _ = M.build_local_aggregation_module(20, EasyDict({'MLPS': [[32, 32]], 'POOL_RADIUS': [0.8], 'NSAMPLE': [12]}))
    assert not layer.to(dev).train()._train_fused_ok()           # nsample not a multiple of 16
    layer, _ = M.build_local_aggregation_module(20, EasyDict({'MLPS': [[32, 32]], 'POOL_RADIUS': [0.8], 'NSAMPLE': [16]}))
    layer = layer.to(dev).train()
    assert layer._train_fused_ok()
    layer.mlps[0][1].eval()
    assert not layer._train_fused_ok()                           # an eval-mode BatchNorm inside a training module: rows path
