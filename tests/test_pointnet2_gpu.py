"""GPU parity: PointNet++ stack ops, points-in-boxes, RoI-aware pool (HIP through the C-ABI / pcdet.ops mirrors) vs the
oracle. Index outputs bit-exact; gathered features bit-exact; atomically accumulated gradients within 1e-5 relative."""
import numpy as np
import pytest
import torch

import oracle
from boxes_synth import detection_boxes
from synth import kitti_batch

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_ball_query_and_grouping(dev):
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    pts, off, _ = kitti_batch(0, 3, n_points=6000)
    xyz = np.ascontiguousarray(pts[:, :3])
    xc = np.diff(off).astype(np.int32)
    rng = np.random.default_rng(0)
    new = np.concatenate([xyz[off[b]:off[b + 1]][rng.choice(6000, 500, replace=False)] for b in range(3)])
    new[::17] += 50.0                       # some empty balls
    nc = np.array([500, 500, 500], np.int32)
    feat = rng.normal(size=(len(xyz), 19)).astype(np.float32)
    for radius, ns in ((0.4, 16), (0.8, 32), (4.8, 16)):
        idx, empty = U.ball_query(radius, ns, _t(xyz, dev), _t(xc, dev), _t(new, dev), _t(nc, dev))
        ref = oracle.ball_query(radius, ns, xyz, xc, new, nc)
        ref_empty = ref[:, 0] == -1
        ref[ref_empty] = 0
        np.testing.assert_array_equal(idx.cpu().numpy(), ref)
        np.testing.assert_array_equal(empty.cpu().numpy(), ref_empty)
        assert ref_empty.sum() >= 80
        f = _t(feat, dev).requires_grad_(True)
        g = U.grouping_operation(f, _t(xc, dev), idx, _t(nc, dev))
        np.testing.assert_array_equal(g.detach().cpu().numpy(), oracle.group_points(feat, xc, ref, nc))
        go = rng.normal(size=tuple(g.shape)).astype(np.float32)
        g.backward(_t(go, dev))
        # float atomics: the order of the (up to a few hundred) addends per source point differs from the oracle's
        np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.group_points_grad(go, ref, nc, xc, len(xyz)),
                                   rtol=1e-4, atol=1e-4)
    qg = U.QueryAndGroup(0.8, 16, use_xyz=True)
    nf, idx = qg(_t(xyz, dev), _t(xc, dev), _t(new, dev), _t(nc, dev), _t(feat, dev))
    assert nf.shape == (1500, 22, 16)
    assert float(nf[::17].abs().sum()) == 0.0


def test_ball_query_pair_equals_two_queries(dev):
    """crb_ball_query2_stack (both radii in one scan) == two crb_ball_query_stack calls + the reference's empty fix-up"""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    pts, off, _ = kitti_batch(3, 3, n_points=7000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(9)
    counts = (397, 3, 399)        # ragged: groups of 8 queries straddle frame boundaries, the last group is short
    _check_ball_query_pair(dev, xyz, xc, off, rng, (402, 0, 397))     # a frame without queries
    _check_ball_query_pair(dev, xyz, xc, off, rng, counts)


def _check_ball_query_pair(dev, xyz, xc, off, rng, counts):
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    new = torch.cat([xyz[off[b]:off[b + 1]][torch.from_numpy(rng.choice(7000, c, replace=False)).to(dev)]
                     for b, c in enumerate(counts)]).contiguous()
    new[::11] += 60.0
    nc = torch.tensor(counts, dtype=torch.int32, device=dev)
    for (ra, na, rb, nb) in ((0.4, 16, 0.8, 16), (2.4, 16, 4.8, 32), (1.6, 24, 0.8, 7), (0.9, 5, 0.9, 64)):
        (ia, ea), (ib, eb) = U.ball_query_pair(ra, na, rb, nb, xyz, xc, new, nc)
        ja, fa = U.ball_query(ra, na, xyz, xc, new, nc)
        jb, fb = U.ball_query(rb, nb, xyz, xc, new, nc)
        assert torch.equal(ia, ja) and torch.equal(ib, jb)
        assert torch.equal(ea.bool(), fa) and torch.equal(eb.bool(), fb)
        assert int(fa.sum()) >= 60
        for r, n, got_i, got_e in ((ra, na, ia, ea), (rb, nb, ib, eb)):          # and the oracle (scan order, first-hit padding)
            ref = oracle.ball_query(r, n, xyz.cpu().numpy(), xc.cpu().numpy(), new.cpu().numpy(),
                                    np.asarray(counts, dtype=np.int32))
            ref_empty = ref[:, 0] == -1
            ref[ref_empty] = 0
            np.testing.assert_array_equal(got_i.cpu().numpy(), ref)
            np.testing.assert_array_equal(got_e.cpu().numpy().astype(bool), ref_empty)


@pytest.mark.parametrize('radii', [(0.4, 16, 0.8, 16), (0.8, 16, 1.2, 32), (1.2, 16, 2.4, 32), (0.05, 4, 0.1, 8), (0.8, 16, 0.8, 16)])
def test_grid_ball_query_equals_the_scan_at_the_bench_size(dev, monkeypatch, radii):
    """crb_ball_query2_grid_stack (counting sort of the call's points into cells of the larger radius, 27 cells per query, the
    nsample smallest indices kept in ascending order) against crb_ball_query2_stack (scan of the whole frame) on 4 frames of
    20,000 lidar points with 2,048 sampled + 200 far-away queries per frame, duplicated points included: index lists and empty
    flags equal. Dense balls (near the sensor: hundreds of hits, more than 1,024 candidates at the large radii) take the kernel's
    scan fallback, sparse ones the selection path; both are compared here, and against the oracle in the test above."""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    ra, na, rb, nb = radii
    B, n = 4, 20000
    pts, off, _ = kitti_batch(11, B, n_points=n)
    xyz_np = np.ascontiguousarray(pts[:, :3]).copy()
    xyz_np[5:4000:7] = xyz_np[4:3999:7]                                   # exact duplicates (equal distances, different indices)
    xyz = _t(xyz_np, dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(5)
    new = torch.cat([torch.cat([xyz[off[b]:off[b + 1]][torch.from_numpy(rng.choice(n, 2048, replace=False)).to(dev)],
                                xyz[off[b]:off[b] + 200] + 300.0]) for b in range(B)]).contiguous()
    nc = torch.full((B,), 2248, dtype=torch.int32, device=dev)
    monkeypatch.setattr(U, 'BALL_QUERY_GRID', True)
    (ia, ea), (ib, eb) = U.ball_query_pair(ra, na, rb, nb, xyz, xc, new, nc)
    monkeypatch.setattr(U, 'BALL_QUERY_GRID', False)
    (ja, fa), (jb, fb) = U.ball_query_pair(ra, na, rb, nb, xyz, xc, new, nc)
    assert torch.equal(ea, fa) and torch.equal(eb, fb) and int(fb.sum()) >= 200 * B
    assert torch.equal(ia, ja) and torch.equal(ib, jb)


@pytest.mark.parametrize('n,m', [(20000, 2048), (4096, 512), (1000, 100), (777, 64), (37, 10), (30000, 300), (50000, 200)])
def test_fps(dev, n, m):
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    pts, off, _ = kitti_batch(5, 2, n_points=n)
    xyz = np.ascontiguousarray(pts[:, :3].reshape(2, n, 3))
    out = U.farthest_point_sample(_t(xyz, dev), m)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.fps(xyz, m))


def test_fps_tie_rule_with_duplicate_points(dev):
    """duplicated points create exact distance ties: the HIP kernel must resolve them like the reference's LDS tree"""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    rng = np.random.default_rng(4)
    base = rng.normal(size=(300, 3)).astype(np.float32)
    xyz = np.concatenate([base, base, base[:150]])[rng.permutation(750)][None]
    out = U.farthest_point_sample(_t(xyz, dev), 200)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.fps(xyz, 200))
    xyz2 = np.concatenate([base] * 8)[None]          # n = 2400 > 1024 threads, every point 8 times
    out2 = U.farthest_point_sample(_t(xyz2, dev), 64)
    np.testing.assert_array_equal(out2.cpu().numpy(), oracle.fps(xyz2, 64))


def test_three_nn_and_interpolate(dev):
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    rng = np.random.default_rng(6)
    unknown = rng.uniform(-10, 10, (3000, 3)).astype(np.float32)
    known = rng.uniform(-10, 10, (700, 3)).astype(np.float32)
    uc, kc = np.array([1000, 2000], np.int32), np.array([300, 400], np.int32)
    dist, idx = U.three_nn(_t(unknown, dev), _t(uc, dev), _t(known, dev), _t(kc, dev))
    d2, ridx = oracle.three_nn(unknown, uc, known, kc)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_allclose(dist.cpu().numpy(), np.sqrt(d2), rtol=1e-6)
    feat = rng.normal(size=(700, 33)).astype(np.float32)
    w = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    f = _t(feat, dev).requires_grad_(True)
    out = U.three_interpolate(f, idx, _t(w, dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), oracle.three_interpolate(feat, ridx, w), rtol=1e-6, atol=1e-6)
    go = rng.normal(size=(3000, 33)).astype(np.float32)
    out.backward(_t(go, dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(go, ridx, w, 700), rtol=1e-4,
                               atol=1e-4)


def test_points_in_boxes(dev):
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils as R
    pts, off, gt = kitti_batch(3, 4)
    xyz = np.ascontiguousarray(pts[:, :3].reshape(4, 20000, 3))
    rng = np.random.default_rng(1)
    boxes = np.stack([np.concatenate([gt[b, :, :7], detection_boxes(rng, 300)[0]]) for b in range(4)])
    got = R.points_in_boxes_gpu(_t(xyz, dev), _t(boxes, dev)).cpu().numpy()
    ref = oracle.points_in_boxes(xyz, boxes)
    np.testing.assert_array_equal(got, ref)
    assert (ref >= 0).sum() > 1000 and (ref == -1).sum() > 1000
    m = R.points_in_boxes_cpu(xyz[0, :500], boxes[0, :20])
    assert m.shape == (20, 500)
    # empty box list
    e = R.points_in_boxes_gpu(_t(xyz[:1], dev), torch.zeros((1, 0, 7), device=dev))
    assert (e == -1).all()


@pytest.mark.parametrize('method', ['max', 'avg'])
def test_roiaware_pool(dev, method):
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils as R
    pts, off, gt = kitti_batch(9, 1)
    xyz = np.ascontiguousarray(pts[:, :3])
    rng = np.random.default_rng(2)
    rois = gt[0, :, :7].copy()
    rois[:, 3:6] *= 1.5
    feat = rng.normal(size=(len(xyz), 16)).astype(np.float32)
    pool = R.RoIAwarePool3d(out_size=6, max_pts_each_voxel=8)       # small cap: exercises "first max_pts-1 in order"
    f = _t(feat, dev).requires_grad_(True)
    out = pool(_t(rois, dev), _t(xyz, dev), f, pool_method=method)
    ref, argmax, pidx = oracle.roiaware_pool(rois, xyz, feat, (6, 6, 6), 8, 0 if method == 'max' else 1)
    if method == 'max':
        np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)
    else:
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert (pidx[..., 0] == 7).any() and (pidx[..., 0] > 0).sum() > 50
    go = rng.normal(size=tuple(out.shape)).astype(np.float32)
    out.backward(_t(go, dev))
    g = np.zeros_like(feat, dtype=np.float64)
    N = len(rois)
    for bi in range(N):
        for cell in range(216):
            v = pidx[bi].reshape(216, 8)[cell]
            for c in range(16):
                if method == 'max':
                    am = argmax[bi].reshape(216, 16)[cell, c]
                    if am >= 0:
                        g[am, c] += go[bi].reshape(216, 16)[cell, c]
                else:
                    for k in range(1, v[0] + 1):
                        g[v[k], c] += go[bi].reshape(216, 16)[cell, c] / max(v[0], 1)
    np.testing.assert_allclose(f.grad.cpu().numpy(), g, rtol=1e-4, atol=1e-5)


def test_sa_module_eval_bn_folding_equals_module_path(dev):
    from pcdet.config import EasyDict
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    torch.manual_seed(0)
    layer, c_out = M.build_local_aggregation_module(
        12, EasyDict({'MLPS': [[16, 16], [16, 32]], 'POOL_RADIUS': [0.8, 1.6], 'NSAMPLE': [16, 16]}))
    layer = layer.to(dev)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    layer.eval()
    xyz = torch.rand(3000, 3, device=dev) * 10
    feat = torch.randn(3000, 12, device=dev)
    new = xyz[::7].contiguous()
    cnt = torch.tensor([3000], dtype=torch.int32, device=dev)
    ncnt = torch.tensor([new.shape[0]], dtype=torch.int32, device=dev)
    with torch.no_grad():
        _, a = layer(xyz, cnt, new, ncnt, feat)            # folded path
    with torch.enable_grad():
        _, b = layer(xyz, cnt, new, ncnt, feat)            # module path
    assert a.shape == (new.shape[0], c_out)
    torch.testing.assert_close(a, b.detach(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('c_in,mlps,nsamples', [(128, [[64, 64], [64, 64]], [16, 16]), (5, [[32, 32], [16, 64]], [24, 7]),
                                                (1, [[16, 16], [64, 32]], [16, 40])])
def test_fused_sa_eval_kernel_equals_module_path(dev, c_in, mlps, nsamples):
    """crb_sa_mlp2_max_stack (group + MLP + max in one launch) against the Conv2d/BatchNorm2d/max_pool2d module path:
    two frames, empty balls, nsample below / above / not a multiple of the 16-row MFMA tile. fp32, rtol 1e-4."""
    from pcdet.config import EasyDict
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    torch.manual_seed(3)
    layer, c_out = M.build_local_aggregation_module(
        c_in, EasyDict({'MLPS': mlps, 'POOL_RADIUS': [0.8, 1.6], 'NSAMPLE': nsamples}))
    layer = layer.to(dev)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    layer.eval()
    pts, off, _ = kitti_batch(2, 5, n_points=6000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(2)
    sel = np.concatenate([rng.choice(6000, 700, replace=False), 6000 + rng.choice(6000, 401, replace=False)])
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    new[::5] += 55.0                                        # empty balls
    nc = torch.tensor([700, 401], dtype=torch.int32, device=dev)
    feat = torch.randn(12000, c_in, device=dev)
    with torch.no_grad():
        _, a = layer(xyz, xc, new, nc, feat)                # fused kernel
        M.FUSED_SA_EVAL = False
        try:
            _, c = layer(xyz, xc, new, nc, feat)            # folded convs
        finally:
            M.FUSED_SA_EVAL = True
    with torch.enable_grad():
        _, b = layer(xyz, xc, new, nc, feat)                # module path
    assert a.shape == (1101, c_out)
    torch.testing.assert_close(a, b.detach(), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c, b.detach(), rtol=1e-4, atol=2e-5)
    assert float(a[::5].std(dim=0).max()) == 0.0            # empty balls: one constant row


def test_sa_module_rows_training_path_equals_module_path(dev):
    """training: row-major grouping + F.linear + fused BN/ReLU row kernels + max == Conv2d/BatchNorm2d/max_pool2d modules:
    outputs (1e-4), grads w.r.t. features (1e-3) and every parameter (5e-3), running statistics (1e-4). fp32."""
    import copy
    from pcdet.config import EasyDict
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    torch.manual_seed(4)
    layer, c_out = M.build_local_aggregation_module(
        20, EasyDict({'MLPS': [[32, 32], [64, 64]], 'POOL_RADIUS': [0.8, 1.6], 'NSAMPLE': [16, 16]}))
    layer = layer.to(dev).train()
    ref = copy.deepcopy(layer)
    pts, off, _ = kitti_batch(2, 7, n_points=6000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(2)
    sel = np.concatenate([rng.choice(6000, 500, replace=False), 6000 + rng.choice(6000, 300, replace=False)])
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    new[::6] += 55.0
    nc = torch.tensor([500, 300], dtype=torch.int32, device=dev)
    f1 = torch.randn(12000, 20, device=dev, requires_grad=True)
    f2 = f1.detach().clone().requires_grad_(True)
    go = torch.randn(800, c_out, device=dev)
    assert M.ROWS_TRAIN
    M.FUSED_TRAIN, fused = False, M.FUSED_TRAIN             # this test pins the rows path; the fused node has its own below
    try:
        _, a = layer(xyz, xc, new, nc, f1)
        a.backward(go)
        M.ROWS_TRAIN = False
        _, b = ref(xyz, xc, new, nc, f2)
        b.backward(go)
    finally:
        M.ROWS_TRAIN = True
        M.FUSED_TRAIN = fused
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)

    def same_up_to_relu_flips(g1, g2, what):
        # sums over M*ns = 12.8k rows in another association (GEMM vs conv, BN reductions); a pre-activation within rounding
        # of 0 may also land on different sides of the ReLU in the two paths and change a handful of entries
        d, scale = (g1 - g2).abs(), max(1.0, float(g2.abs().max()))
        assert float(d.median()) < 1e-3 * scale, what
        assert float((d > 5e-3 * scale).float().mean()) < 0.02, what
        assert float(d.norm() / g2.norm().clamp_min(1e-12)) < 2e-2, what
    same_up_to_relu_flips(f1.grad, f2.grad, 'feature grad')
    for (n1, p1), (n2, p2) in zip(layer.named_parameters(), ref.named_parameters()):
        same_up_to_relu_flips(p1.grad, p2.grad, n1)
    for (n1, b1), (n2, b2) in zip(layer.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=lambda m, n1=n1: n1 + ': ' + m)


@pytest.mark.parametrize('mlps,nsample', [([[32, 32], [64, 64]], [16, 16]), ([[16, 16], [16, 32]], [16, 32]),
                                          ([[64, 32], [32, 64]], [32, 16])])
def test_sa_module_fused_training_node_equals_module_and_rows_paths(dev, mlps, nsample):
    """training, two-layer MLPs: the recompute node (csrc/sa_mlp_train.hip: no (M*ns, H) activation kept) against the
    Conv2d / BatchNorm2d / max_pool2d modules (outputs 1e-4, gradients up to ReLU flips, running statistics 1e-4) and against the
    rows path (same arithmetic for layer 1 and the BatchNorms, MFMA instead of rocBLAS for layer 2: outputs 2e-5); bit-equal
    re-runs; empty balls, two frames, nsample 16 and 32."""
    import copy
    from pcdet.config import EasyDict
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M
    torch.manual_seed(4)
    layer, c_out = M.build_local_aggregation_module(20, EasyDict({'MLPS': mlps, 'POOL_RADIUS': [0.8, 1.6], 'NSAMPLE': nsample}))
    layer = layer.to(dev).train()
    with torch.no_grad():                                   # BatchNorm parameters away from (1, 0), some gammas negative
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.randn_like(m.weight) * 0.5 + 0.8)
                m.bias.copy_(torch.randn_like(m.bias) * 0.3)
    ref, rows, again = copy.deepcopy(layer), copy.deepcopy(layer), copy.deepcopy(layer)
    pts, off, _ = kitti_batch(2, 7, n_points=6000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(2)
    sel = np.concatenate([rng.choice(6000, 500, replace=False), 6000 + rng.choice(6000, 301, replace=False)])
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    new[::6] += 55.0
    nc = torch.tensor([500, 301], dtype=torch.int32, device=dev)
    feats = [torch.randn(12000, 20, device=dev).requires_grad_(True)]
    feats += [feats[0].detach().clone().requires_grad_(True) for _ in range(3)]
    go = torch.randn(801, c_out, device=dev)
    assert M.ROWS_TRAIN and M.FUSED_TRAIN and layer._train_fused_ok()
    _, a = layer(xyz, xc, new, nc, feats[0])
    assert type(a.grad_fn).__name__ == 'SAMlp2TrainConcatBackward'
    a.backward(go)
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    floor = U.SORTED_SCATTER_MIN_PAIRS
    try:                                                    # the second run scatters in source-row order (the large layers' path)
        U.SORTED_SCATTER_MIN_PAIRS = 0
        assert U.SORTED_SCATTER
        _, a2 = again(xyz, xc, new, nc, feats[3])
        a2.backward(go)
    finally:
        U.SORTED_SCATTER_MIN_PAIRS = floor
    try:
        M.FUSED_TRAIN = False
        _, r = rows(xyz, xc, new, nc, feats[2])
        r.backward(go)
        M.ROWS_TRAIN = False
        _, b = ref(xyz, xc, new, nc, feats[1])
        b.backward(go)
    finally:
        M.ROWS_TRAIN = True
        M.FUSED_TRAIN = True
    # re-run: outputs, BatchNorm gradients and the second conv's weight gradient come from ordered partial sums (bit-equal); the
    # feature gradient and the first conv's weight gradient pass through the float atomics of the scatter into dP (as on the rows
    # path) and agree to rounding
    assert torch.equal(a, a2)
    for (n1, p1), (n2, p2) in zip(layer.named_parameters(), again.named_parameters()):
        if n1.endswith('.0.weight'):
            torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-5, atol=1e-5 * float(p2.grad.abs().max()))
        else:
            assert torch.equal(p1.grad, p2.grad), n1
    torch.testing.assert_close(feats[0].grad, feats[3].grad, rtol=1e-5, atol=1e-5 * float(feats[3].grad.abs().max()))
    torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(a, r, rtol=2e-5, atol=2e-5)

    def same_up_to_relu_flips(g1, g2, what):
        d, scale = (g1 - g2).abs(), max(1.0, float(g2.abs().max()))
        assert float(d.median()) < 1e-3 * scale, what
        assert float((d > 5e-3 * scale).float().mean()) < 0.02, what
        assert float(d.norm() / g2.norm().clamp_min(1e-12)) < 2e-2, what
    for other, fo, tag in ((ref, feats[1], 'modules'), (rows, feats[2], 'rows')):
        same_up_to_relu_flips(feats[0].grad, fo.grad, tag + ': feature grad')
        for (n1, p1), (n2, p2) in zip(layer.named_parameters(), other.named_parameters()):
            same_up_to_relu_flips(p1.grad, p2.grad, tag + ': ' + n1)
        for (n1, b1), (n2, b2) in zip(layer.named_buffers(), other.named_buffers()):
            torch.testing.assert_close(b1.float(), b2.float(), rtol=1e-4, atol=1e-5, msg=lambda m, n1=n1: tag + ' ' + n1 + ': ' + m)


def test_grouped_first_layer_rows_equals_group_then_gemm(dev):
    """y = [xyz_j - c_i ; f_j] W^T taken as gather(F W1f^T) + W1x (xyz_j - c_i) (no grouped matrix) == grouping followed by the
    GEMM: values 1e-5, feature / weight gradients 1e-4 relative to their scale (other summation order, fp32)."""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    pts, off, _ = kitti_batch(3, 3, n_points=5000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(5)
    sel = np.concatenate([k * 5000 + rng.choice(5000, n, replace=False) for k, n in enumerate((310, 5, 202))])
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    new[::7] += 45.0                                         # empty balls
    nc = torch.tensor([310, 5, 202], dtype=torch.int32, device=dev)
    torch.manual_seed(0)
    for C, H, ns in ((21, 16, 16), (128, 64, 16), (7, 32, 5), (9, 128, 16)):
        f1 = torch.randn(15000, C, device=dev, requires_grad=True)
        f2 = f1.detach().clone().requires_grad_(True)
        w1 = (torch.randn(H, 3 + C, device=dev) * 0.2).requires_grad_(True)
        w2 = w1.detach().clone().requires_grad_(True)
        ball = U.ball_query(1.1, ns, xyz, xc, new, nc)
        assert bool(ball[1].any()) and not bool(ball[1].all())
        a = U.grouped_first_layer_rows(1.1, ns, xyz, xc, new, nc, f1, w1, ball=ball)
        rows, _ = U.query_and_group_rows(1.1, ns, xyz, xc, new, nc, f2, ball=ball)
        b = rows @ w2.t()
        assert a.shape == b.shape == (517 * ns, H)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
        assert float(a.detach().view(517, ns, H)[ball[1]].abs().max()) == 0.0
        go = torch.randn_like(a)
        a.backward(go)
        b.backward(go)
        for g1, g2 in ((f1.grad, f2.grad), (w1.grad, w2.grad)):
            assert float((g1 - g2).abs().max()) < 1e-4 * max(1.0, float(g2.abs().max()))
    with torch.no_grad():                                    # any other width: scalar-lane forward kernel
        f, w = torch.randn(15000, 10, device=dev), torch.randn(24, 13, device=dev) * 0.2
        ball = U.ball_query(1.1, 16, xyz, xc, new, nc)
        a = U.grouped_first_layer_rows(1.1, 16, xyz, xc, new, nc, f, w, ball=ball)
        b = U.query_and_group_rows(1.1, 16, xyz, xc, new, nc, f, ball=ball)[0] @ w.t()
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)


def test_fused_query_group_equals_query_and_group(dev):
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    pts, off, _ = kitti_batch(2, 2, n_points=5000)
    xyz = _t(np.ascontiguousarray(pts[:, :3]), dev)
    xc = _t(np.diff(off).astype(np.int32), dev)
    rng = np.random.default_rng(1)
    sel = np.concatenate([rng.choice(5000, 300, replace=False), 5000 + rng.choice(5000, 200, replace=False)])
    new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
    new[::9] += 40.0
    nc = torch.tensor([300, 200], dtype=torch.int32, device=dev)
    f1 = torch.randn(10000, 21, device=dev, requires_grad=True)
    f2 = f1.detach().clone().requires_grad_(True)
    a, idx_a = U.query_and_group_fused(0.9, 16, xyz, xc, new, nc, f1)
    b, idx_b = U.QueryAndGroup(0.9, 16, use_xyz=True)(xyz, xc, new, nc, f2)
    assert torch.equal(idx_a, idx_b)
    assert torch.equal(a, b.permute(1, 0, 2).unsqueeze(0))
    g = torch.randn_like(a)
    a.backward(g)
    b.backward(g[0].permute(1, 0, 2))
    torch.testing.assert_close(f1.grad, f2.grad, rtol=1e-5, atol=1e-5)


def test_training_sa_ops_full_size_properties(dev):
    """size-independent checks of the training-path SA kernels at the PV-RCNN RoI-grid shape (16 x 128 x 216 queries x 16 samples
    = 7.08 M rows x 64 channels, 1.8 GB per activation), where the CPU oracle would take minutes:
    * gathered first layer: linear in the features and in the weight (rtol 1e-4 of the output scale), and its backward is the
      adjoint of its forward, <y, G f> == <G^T y, f> and the same for the weight (relative 1e-4, float atomics / other order);
    * BN+ReLU+max: invariant to the order of the rows inside a group and to a positive affine map of the input (BatchNorm
      removes it), gradient rows sum to ~0 per channel (BatchNorm's backward projects the mean out)."""
    from crbhip import bnrelu
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    torch.manual_seed(0)
    B, NK, R, G3, C, H, ns = 16, 2048, 128, 216, 128, 64, 16
    xyz = (torch.rand(B * NK, 3, device=dev) * torch.tensor([70.0, 80.0, 4.0], device=dev)).contiguous()
    xc = torch.full((B,), NK, dtype=torch.int32, device=dev)
    centres = xyz.view(B, NK, 3)[:, torch.randint(0, NK, (R,), device=dev)]
    new = (centres[:, :, None, :] + (torch.rand(B, R, G3, 3, device=dev) - 0.5) * 4.0).reshape(-1, 3).contiguous()
    nc = torch.full((B,), R * G3, dtype=torch.int32, device=dev)
    ball = U.ball_query(1.6, ns, xyz, xc, new, nc)
    f1 = torch.randn(B * NK, C, device=dev)
    f2 = torch.randn(B * NK, C, device=dev)
    w1 = torch.randn(H, 3 + C, device=dev) * 0.1
    w2 = torch.randn(H, 3 + C, device=dev) * 0.1
    fwd = lambda f, w: U.grouped_first_layer_rows(1.6, ns, xyz, xc, new, nc, f, w, ball=ball)
    y11 = fwd(f1, w1)
    assert y11.shape == (B * R * G3 * ns, H)
    scale = float(y11.abs().max())
    y0 = fwd(torch.zeros_like(f1), w1)                        # the offset term W1x (xyz_j - c_i): y is affine in f
    d = fwd(2.0 * f1 - 0.5 * f2, w1) - (2.0 * y11 - 0.5 * fwd(f2, w1) - 0.5 * y0)
    assert float(d.abs().max()) < 1e-4 * scale
    d = fwd(f1, w1 + w2) - (y11 + fwd(f1, w2))
    assert float(d.abs().max()) < 1e-4 * scale
    del d
    fa = f1.clone().requires_grad_(True)
    wa = w1.clone().requires_grad_(True)
    y = fwd(fa, wa)
    go = torch.randn_like(y)
    gf, gw = torch.autograd.grad(y, (fa, wa), go)
    lhs = float((go.double() * y.detach().double()).sum())                       # <go, y>, y bilinear in (f | 1, w)
    rhs_w = float((gw.double() * w1.double()).sum())                             # y is linear in w: <go, y> = <gw, w>
    assert abs(lhs - rhs_w) < 1e-4 * max(1.0, abs(lhs))
    rhs_f = float((gf.double() * f1.double()).sum()) + float((go.double() * y0.double()).sum())
    assert abs(lhs - rhs_f) < 1e-4 * max(1.0, abs(lhs))
    del y, gf, gw, fa, wa

    bn = torch.nn.BatchNorm1d(H).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
    x = y11.detach().requires_grad_(True)
    z = bnrelu.bn_relu_max_concat([x], [ns], [bn])
    M = B * R * G3
    assert z.shape == (M, H) and float(z.detach().min()) >= 0.0
    perm = torch.argsort(torch.rand(M, ns, device=dev), dim=1)                   # shuffle the rows inside every group
    xp = torch.gather(y11.view(M, ns, H), 1, perm[..., None].expand(-1, -1, H)).reshape(-1, H).contiguous()
    zp = bnrelu.bn_relu_max_concat([xp], [ns], [bn])
    assert float((zp - z.detach()).abs().max()) < 1e-5 * max(1.0, float(z.detach().abs().max()))  # statistics summed in another order
    za = bnrelu.bn_relu_max_concat([(3.0 * y11 + 0.7).contiguous()], [ns], [bn])
    assert float((za - z.detach()).abs().max()) < 2e-4 * max(1.0, float(z.detach().abs().max()))
    gx, = torch.autograd.grad(z, x, torch.randn_like(z))
    assert float(gx.sum(0).abs().max()) < 1e-3 * float(gx.abs().sum(0).max())


def test_grouped_ball_query_equals_ungrouped(dev):
    """crb_ball_query2_grouped_stack (per-group bounding-sphere prefilter, RoI grid pooling) returns the index lists and empty
    flags of crb_ball_query2_stack bit for bit: compact groups (RoI-like), groups as large as the frame (every point a
    candidate -> the fallback scan when more than 2048 candidates), frames with different numbers of groups, ra > rb"""
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    rng = np.random.default_rng(3)
    seen_empty = seen_full = 0
    for n_pts, group, groups_per_frame, spread in ((2048, 216, (7, 0, 12, 3), 2.0), (6000, 64, (5, 9), 300.0),
                                                  (1500, 216, (4, 4, 4), 1.0), (700, 1000, (1, 2), 30.0)):
        B = len(groups_per_frame)
        xyz = (rng.random((B * n_pts, 3)) * np.array([70.0, 80.0, 4.0])).astype(np.float32)
        xc = np.full((B,), n_pts, np.int32)
        new = []
        for b, g in enumerate(groups_per_frame):
            for _ in range(g):
                c = xyz[b * n_pts + rng.integers(0, n_pts)]
                new.append(c + (rng.random((group, 3)).astype(np.float32) - 0.5) * spread)
        new = np.concatenate(new).astype(np.float32)
        nc = np.array([g * group for g in groups_per_frame], np.int32)
        for (ra, na, rb, nb) in ((0.8, 16, 1.6, 16), (1.6, 16, 0.8, 32), (2.4, 5, 2.4, 64)):
            args = (ra, na, rb, nb, _t(xyz, dev), _t(xc, dev), _t(new, dev), _t(nc, dev))
            (ia, ea), (ib, eb) = U.ball_query_pair(*args)
            (ja, fa), (jb, fb) = U.ball_query_pair(*args, group=group)
            assert torch.equal(ia, ja) and torch.equal(ib, jb) and torch.equal(ea, fa) and torch.equal(eb, fb)
            seen_empty += int(ea.sum())
            seen_full += int((ea == 0).sum())
    assert seen_empty > 100 and seen_full > 100
    # a caller that breaks the "no group straddles a frame" promise still gets the ungrouped result (slow path)
    xyz = (rng.random((2 * 900, 3)) * np.array([30.0, 30.0, 4.0])).astype(np.float32)
    new = xyz[rng.choice(1800, 432, replace=False)] + 0.1
    for counts in ((100, 332), (300, 132)):
        new_s = np.concatenate([new[:counts[0]] % np.array([30, 30, 4], np.float32), new[counts[0]:]]).astype(np.float32)
        args = (0.9, 16, 1.8, 16, _t(xyz, dev), _t(np.array([900, 900], np.int32), dev), _t(new_s, dev),
                _t(np.array(counts, np.int32), dev))
        (ia, ea), (ib, eb) = U.ball_query_pair(*args)
        (ja, fa), (jb, fb) = U.ball_query_pair(*args, group=216)
        assert torch.equal(ia, ja) and torch.equal(ib, jb) and torch.equal(ea, fa) and torch.equal(eb, fb)


def test_first_layer_bn_relu_as_one_node_equals_the_separate_ops(dev):
    """StackSAModuleMSG in training mode with FUSED_FIRST_BN (first conv + BatchNorm + ReLU as one autograd node, the BatchNorm
    backward applied inside crb_group_affine_rows_grad_bn_stack) against the same module with the separate ops: outputs and
    running statistics bit-identical (the forward kernels are the same), every gradient to 1e-5 of its largest entry (the
    scatter into the per-source-point gradient uses float atomics in both paths: last-bit run-to-run differences)."""
    import copy
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as pm
    torch.manual_seed(5)
    B, n, m, C = 2, 3000, 700, 16
    xyz = torch.rand(B * n, 3, device=dev) * torch.tensor([20.0, 20.0, 3.0], device=dev)
    new_xyz = xyz.view(B, n, 3)[:, :m].reshape(-1, 3).contiguous() + 0.05
    cnt, ncnt = torch.full((B,), n, dtype=torch.int32, device=dev), torch.full((B,), m, dtype=torch.int32, device=dev)
    feats = torch.randn(B * n, C, device=dev)
    mod_a = pm.StackSAModuleMSG(radii=[0.8, 1.6], nsamples=[16, 16], mlps=[[C, 32, 32], [C, 64, 64]], use_xyz=True,
                                pool_method='max_pool').to(dev).train()
    mod_b, mod_b0 = copy.deepcopy(mod_a), copy.deepcopy(mod_a)
    gout = torch.randn(B * m, 96, device=dev)

    def run(mod, fused):
        old = pm.FUSED_FIRST_BN
        pm.FUSED_FIRST_BN = fused
        try:
            f = feats.clone().requires_grad_(True)
            _, out = mod(xyz, cnt, new_xyz, ncnt, f)
            out.backward(gout)
        finally:
            pm.FUSED_FIRST_BN = old
        return out.detach(), f.grad, {k: p.grad for k, p in mod.named_parameters()}, {k: v.clone() for k, v in mod.named_buffers()}
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu
    ob, fb, pb, bb = run(mod_b, False)
    pu.FIRST_LAYER_SLAB_STATS = False            # same statistics kernels as the separate ops: bit-identical forward
    try:
        oa, fa, pa, ba = run(copy.deepcopy(mod_b0), True)
    finally:
        pu.FIRST_LAYER_SLAB_STATS = True
    assert torch.equal(oa, ob)
    for k in bb:
        assert torch.equal(ba[k], bb[k]), k
    oa, fa, pa, ba = run(mod_a, True)            # statistics from the producer's slab sums: another summation order
    def close(x, y, what):
        assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max()) + 1e-9, (what, float((x - y).abs().max()), float(y.abs().max()))
    close(oa, ob, 'output')
    for k in bb:
        close(ba[k].float(), bb[k].float(), k)
    close(fa, fb, 'features')
    for k in pb:
        assert pa[k] is not None, k
        close(pa[k], pb[k], k)


@pytest.mark.gpu
def test_bev_interpolation_kernel_equals_the_torch_expression(dev):
    """crb_bev_interpolate_forward / _backward against the torch restatement of interpolate_from_bev_features it replaces (itself
    pinned by tests/golden/ref_glue.npz): forward bit-equal (same operations, same order), map gradient to atomic-order rounding;
    keypoints on and outside the map border exercise the clamped corners"""
    from pcdet.models.backbones_3d.pfe import voxel_set_abstraction as vsa
    torch.manual_seed(3)
    B, C, H, W, M = 3, 64, 25, 22, 4000
    bev = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rng = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
    kp = torch.empty(M, 4, device=dev)
    kp[:, 0] = torch.randint(0, B, (M,), device=dev).float()
    kp[:, 1] = torch.rand(M, device=dev) * 74.0 - 2.0          # some left / right of the map
    kp[:, 2] = torch.rand(M, device=dev) * 84.0 - 42.0
    kp[:, 3] = 0.0
    kp[:8, 1] = torch.tensor([0.0, 70.4, 70.39, 0.4, 3.2, 35.2, 70.4, 0.0], device=dev)
    kp[:8, 2] = torch.tensor([-40.0, 40.0, 39.99, -39.6, 0.0, 0.0, -40.0, 40.0], device=dev)

    class Host(object):
        point_cloud_range, voxel_size = rng, [0.05, 0.05, 0.1]
    g = torch.randn(M, C, device=dev)
    outs = {}
    for flag in (False, True):
        vsa.BEV_INTERP_KERNEL = flag
        try:
            bev.grad = None
            out = vsa.VoxelSetAbstraction.interpolate_from_bev_features(Host(), kp, bev, B, 8 * (70.4 / 0.05 / 8 / W))
            (out * g).sum().backward()
            outs[flag] = (out.detach().clone(), bev.grad.clone())
        finally:
            vsa.BEV_INTERP_KERNEL = True
    assert torch.equal(outs[True][0], outs[False][0])
    err = float((outs[True][1] - outs[False][1]).abs().max() / outs[False][1].abs().max())
    assert err < 1e-5, err


@pytest.mark.gpu
def test_bev_interpolation_ignores_rows_of_no_frame_and_honours_deterministic_mode(dev):
    """a keypoint row whose frame index is outside [0, B) (a padded row) reads nothing - its output is zero - and adds
    nothing to the map gradient (the round-4 kernel indexed the map with it); under torch.use_deterministic_algorithms the
    sort-based torch path runs (bit-reproducible map gradient) instead of the kernel's float atomics"""
    from pcdet.models.backbones_3d.pfe import voxel_set_abstraction as vsa
    torch.manual_seed(5)
    B, C, H, W, M = 2, 32, 25, 22, 600
    bev = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    kp = torch.empty(M, 4, device=dev)
    kp[:, 0] = torch.randint(0, B, (M,), device=dev).float()
    kp[:, 1] = torch.rand(M, device=dev) * 70.4
    kp[:, 2] = torch.rand(M, device=dev) * 80.0 - 40.0
    kp[:, 3] = 0.0
    bad = torch.tensor([3, 77, 401], device=dev)
    kp2 = kp.clone()
    kp2[bad, 0] = torch.tensor([-1.0, float(B), 1000.0], device=dev)

    class Host(object):
        point_cloud_range, voxel_size = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1]
    stride = 8 * (70.4 / 0.05 / 8 / W)
    g = torch.randn(M, C, device=dev)
    out = vsa.VoxelSetAbstraction.interpolate_from_bev_features(Host(), kp2, bev, B, stride)
    (out * g).sum().backward()
    grad_bad = bev.grad.clone()
    assert float(out[bad].abs().max()) == 0.0
    keep = torch.ones(M, dtype=torch.bool, device=dev)
    keep[bad] = False
    bev.grad = None
    out_ok = vsa.VoxelSetAbstraction.interpolate_from_bev_features(Host(), kp[keep], bev, B, stride)
    (out_ok * g[keep]).sum().backward()
    assert torch.equal(out[keep], out_ok)
    assert float((grad_bad - bev.grad).abs().max()) <= 1e-5 * float(bev.grad.abs().max())
    # deterministic mode: the torch path, twice the same bits
    grads = []
    torch.use_deterministic_algorithms(True)
    try:
        for _ in range(2):
            bev.grad = None
            o = vsa.VoxelSetAbstraction.interpolate_from_bev_features(Host(), kp, bev, B, stride)
            assert not isinstance(o.grad_fn, vsa._BevInterpolate._backward_cls)
            (o * g).sum().backward()
            grads.append(bev.grad.clone())
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(grads[0], grads[1])


def test_rows_per_frame_from_the_sorted_frame_column(dev):
    """crb_sorted_key_counts (common_utils.batch_counts on device tensors) against numpy's bincount: frames without rows at the
    start, in the middle and at the end, a float column of a wider row-major tensor (strided, as points[:, 0]), an int32 column of
    voxel coordinates, int64 keys, one row, no rows, counts above 64^3 rows"""
    from pcdet.utils import common_utils
    rng = np.random.default_rng(11)
    cases = [[0, 5, 0, 0, 7, 1, 0], [3], [0], [0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [20000] * 16,
             [int(v) for v in rng.integers(0, 40000, size=16)], [300000, 0, 1], [64, 64, 63, 65, 4096, 4097, 1, 0, 262144, 262145]]
    for counts in cases:
        B, n = len(counts), int(sum(counts))
        key = np.repeat(np.arange(B), counts)
        want = np.asarray(counts, dtype=np.int32)
        pts = torch.zeros((n, 5), device=dev)
        pts[:, 0] = torch.from_numpy(key).to(dev).float()
        got = common_utils.batch_counts(pts[:, 0], B)
        assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), want), counts
        coords = torch.zeros((n, 4), dtype=torch.int32, device=dev)
        coords[:, 0] = torch.from_numpy(key).to(dev).int()
        assert np.array_equal(common_utils.batch_counts(coords[:, 0], B).cpu().numpy(), want), counts
        assert np.array_equal(common_utils.batch_counts(torch.from_numpy(key).to(dev), B).cpu().numpy(), want), counts
        # the host form (not the device path) gives the same counts
        assert np.array_equal(common_utils.batch_counts(torch.from_numpy(key), B).numpy(), want), counts
    # frames beyond B are not counted; B larger than the keys present
    key = torch.tensor([0, 0, 1, 3, 3, 3, 5], dtype=torch.int32, device=dev)
    assert common_utils.batch_counts(key, 4).cpu().tolist() == [2, 1, 0, 3]
    assert common_utils.batch_counts(key, 8).cpu().tolist() == [2, 1, 0, 3, 0, 1, 0, 0]


def test_voxel_centers_kernel_equals_the_torch_expression(dev):
    """crb_voxel_centers on the column slice of an (n,4) index tensor against flip + cast + (c + 0.5) * (voxel_size * stride) + minimum:
    bit-equal for every level's stride, n = 0, contiguous (n,3) input"""
    from pcdet.utils import common_utils as CU
    rng = np.random.default_rng(2)
    vs, rng_pc = [0.05, 0.05, 0.1], [0, -40, -3, 70.4, 40, 1]
    for n in (0, 1, 70001):
        idx = torch.from_numpy(np.concatenate([rng.integers(0, 16, (n, 1)), rng.integers(0, 41, (n, 1)), rng.integers(0, 1600, (n, 1)),
                                               rng.integers(0, 1408, (n, 1))], 1).astype(np.int32)).to(dev)
        for times in (1, 2, 4, 8):
            got = CU.get_voxel_centers(idx[:, 1:4], times, vs, rng_pc)
            c = idx[:, 1:4].flip(1).float()
            want = (c + 0.5) * (CU.device_constant(vs, dev) * times) + CU.device_constant(rng_pc[0:3], dev)
            assert got.shape == (n, 3) and torch.equal(got, want)
            assert torch.equal(CU.get_voxel_centers(idx[:, 1:4].contiguous(), times, vs, rng_pc), want)
