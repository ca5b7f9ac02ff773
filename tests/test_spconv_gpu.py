"""GPU parity: HIP rulebooks + gather-GEMM (through the C-ABI / spconv mirror) vs the oracle.
Bit-exact: neighbour tables, output coordinate sets, pair lists (as sorted sets).
fp32 tolerance for features / grads: |hip - oracle| <= 1e-4*|oracle| + 1e-5*max|oracle| (oracle accumulates in double,
the MFMA path is an f32 fma chain over <= 27*128 terms)."""
import numpy as np
import pytest
import torch

import oracle
from synth import random_sparse_coords, kitti_batch, KITTI_RANGE, KITTI_VOXEL

pytestmark = pytest.mark.gpu


def _close(a, ref, rtol=1e-4):
    atol = 1e-5 * max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(a, ref, rtol=rtol, atol=atol)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize('clustered', [True, False])
def test_subm_rulebook_exact(dev, clustered):
    from crbhip import sparse
    rng = np.random.default_rng(10)
    shape = [41, 200, 176]
    coords = random_sparse_coords(rng, 30000, 3, shape, clustered)
    rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), oracle.subm_nbr(coords, shape, [3, 3, 3]))
    pin, pout, pstart = [x.cpu().numpy() for x in rb.pairs()[:3]]
    nbr = rb.nbr.cpu().numpy()
    P = int(pstart[-1])
    assert P == (nbr >= 0).sum()
    for o in (0, 13, 26):
        i = np.nonzero(nbr[:, o] >= 0)[0]
        np.testing.assert_array_equal(pout[pstart[o]:pstart[o + 1]], i)
        np.testing.assert_array_equal(pin[pstart[o]:pstart[o + 1]], nbr[i, o])


@pytest.mark.parametrize('n', [300, 140000])
def test_site_hash_survives_inputs_whose_sites_share_one_x_residue(dev, n):
    """the x-grouped site hash keeps one sub-table per key % 8; with W % 8 == 0 a wall at x = 0 (mod 8) puts every
    key into ONE of them. 300 sites used to overflow the 128 slots a 1,024-slot table gave a residue class (endless probe =
    GPU hang); 140,000 sites (> 128 Ki: the table is sized 2n overall, the class holds 65,536) exercise the spill into the
    next class. Tables stay bit-exact (planned chain and single-table route)."""
    from crbhip import sparse
    rng = np.random.default_rng(77)
    shape = [41, 1600, 1408]
    z = rng.integers(0, shape[0], 3 * n)
    y = rng.integers(0, shape[1], 3 * n)
    x = 8 * rng.integers(0, 6, 3 * n)                       # 6 planes, all x = 0 (mod 8): neighbours in y / z only
    lin = np.unique((z * shape[1] + y) * shape[2] + x)[:n]
    rng.shuffle(lin)
    coords = np.stack([np.zeros_like(lin), lin // (shape[1] * shape[2]), (lin // shape[2]) % shape[1], lin % shape[2]], 1).astype(np.int32)
    assert len(coords) == n and (coords[:, 3] % 8 == 0).all()
    rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), oracle.subm_nbr(coords, shape, [3, 3, 3]))


@pytest.mark.parametrize('ks,st,pd', [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                      ((3, 1, 1), (2, 1, 1), (0, 0, 0))])
def test_spconv_rulebook_exact(dev, ks, st, pd):
    from crbhip import sparse
    rng = np.random.default_rng(11)
    shape = [21, 160, 140]
    coords = random_sparse_coords(rng, 20000, 4, shape)
    rb = sparse.spconv_rulebook(_t(coords, dev), shape, 4, ks, st, pd)
    oc, oshape = oracle.spconv_out(coords, shape, ks, st, pd)
    assert rb.out_shape == oshape
    np.testing.assert_array_equal(rb.out_coords.cpu().numpy(), oc)
    nbr_ref = oracle.spconv_nbr(coords, shape, oc, ks, st, pd)
    np.testing.assert_array_equal(rb.nbr.cpu().numpy(), nbr_ref)
    # transposed table is the exact transpose
    nbr_t = rb.nbr_t.cpu().numpy()
    i, o = np.nonzero(nbr_ref >= 0)
    ref_t = np.full_like(nbr_t, -1)
    ref_t[nbr_ref[i, o], o] = i
    np.testing.assert_array_equal(nbr_t, ref_t)


CHANNELS = [(4, 16), (5, 16), (4, 64), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (32, 16),
            (64, 32), (128, 64)]


@pytest.mark.parametrize('cin,cout', CHANNELS)
def test_subm_conv_fwd_bwd(dev, cin, cout):
    from crbhip import sparse
    rng = np.random.default_rng(100 + cin * 7 + cout)
    shape = [21, 100, 88]
    coords = random_sparse_coords(rng, 5000, 2, shape)
    n = len(coords)
    X = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dY = rng.normal(size=(n, cout)).astype(np.float32)
    rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
    x = _t(X, dev).requires_grad_(True)
    w = _t(W, dev).requires_grad_(True)
    y = sparse.sparse_conv(x, w, rb)
    y.backward(_t(dY, dev))
    nbr = oracle.subm_nbr(coords, shape, [3, 3, 3])
    _close(y.detach().cpu().numpy(), oracle.conv_fwd(X, W, nbr))
    _close(x.grad.cpu().numpy(), oracle.conv_dgrad(dY, W, nbr, n))
    _close(w.grad.cpu().numpy(), oracle.conv_wgrad(X, dY, nbr, 27))


@pytest.mark.parametrize('cin,cout,ks,st,pd', [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                                (64, 64, (3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                                (64, 128, (3, 1, 1), (2, 1, 1), (0, 0, 0))])
def test_strided_conv_fwd_bwd(dev, cin, cout, ks, st, pd):
    from crbhip import sparse
    rng = np.random.default_rng(200 + cin + cout)
    shape = [11, 100, 88]
    coords = random_sparse_coords(rng, 6000, 2, shape)
    n = len(coords)
    K = ks[0] * ks[1] * ks[2]
    X = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(K, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    rb = sparse.spconv_rulebook(_t(coords, dev), shape, 2, ks, st, pd)
    dY = rng.normal(size=(rb.n_out, cout)).astype(np.float32)
    x = _t(X, dev).requires_grad_(True)
    w = _t(W, dev).requires_grad_(True)
    y = sparse.sparse_conv(x, w, rb)
    y.backward(_t(dY, dev))
    oc, _ = oracle.spconv_out(coords, shape, ks, st, pd)
    nbr = oracle.spconv_nbr(coords, shape, oc, ks, st, pd)
    _close(y.detach().cpu().numpy(), oracle.conv_fwd(X, W, nbr))
    _close(x.grad.cpu().numpy(), oracle.conv_dgrad(dY, W, nbr, n))
    _close(w.grad.cpu().numpy(), oracle.conv_wgrad(X, dY, nbr, K))


@pytest.mark.parametrize('level,cin,cout', [(1, 16, 16), (2, 32, 32), (3, 64, 64)])
def test_full_size_linearity_and_adjoint_identity(dev, level, cin, cout):
    """BASELINE configs[1] size (16 frames x 20k points, SubM level 1-3 of VoxelBackBone8x: 190-280k active sites), where
    the CPU oracle would take minutes: size-independent properties instead.
      linearity   conv(a x1 + b x2) = a conv(x1) + b conv(x2)
      adjoint     <conv(x), dy> = <x, dgrad(dy)> = <W, wgrad(x, dy)>      (fwd, dgrad and wgrad describe ONE bilinear map)
      row order   the mask-sorted / heaviest-first table and the natural-order table give the same rows"""
    from crbhip import sparse, voxel
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(_t(pts, dev), _t(off, dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5, want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1))]
    for lvl in range(2, level + 1):
        rbs = sparse.spconv_rulebook(coords, shape, 16, *geo[lvl - 2])
        coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
    rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
    n = rb.n_out
    assert n > 150000
    g = torch.Generator(device=dev).manual_seed(level)
    x1 = torch.randn(n, cin, device=dev, generator=g)
    x2 = torch.randn(n, cin, device=dev, generator=g)
    w = (torch.randn(27, cin, cout, device=dev, generator=g) / cin ** 0.5).requires_grad_(True)
    dy = torch.randn(n, cout, device=dev, generator=g)
    y1, y2 = sparse.sparse_conv(x1, w, rb), sparse.sparse_conv(x2, w, rb)
    y12 = sparse.sparse_conv(0.7 * x1 - 1.3 * x2, w, rb)
    torch.testing.assert_close(y12, 0.7 * y1 - 1.3 * y2, rtol=1e-4, atol=1e-4)
    x = x1.clone().requires_grad_(True)
    y = sparse.sparse_conv(x, w, rb)
    y.backward(dy)
    a = float((y.detach().double() * dy.double()).sum())
    b = float((x.detach().double() * x.grad.double()).sum())
    c = float((w.detach().double() * w.grad.double()).sum())
    scale = float(y.detach().double().norm() * dy.double().norm())
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)
    sparse.MASK_SORT = False
    try:
        rb2 = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        y_nat = sparse.sparse_conv(x1, w.detach(), rb2)
    finally:
        sparse.MASK_SORT = True
    assert torch.equal(y_nat, y1.detach())          # same per-row arithmetic, only the row -> workgroup map differs


def test_conv_results_are_bitwise_reproducible(dev):
    """forward, dgrad and wgrad twice on fresh rulebooks of the same input: bit-identical. (The heaviest-first tile order is
    built with atomics and may differ between the two rulebooks; per-row arithmetic does not depend on it. wgrad partials are
    reduced by a fixed plan and a fixed-shape tree.)"""
    from crbhip import sparse
    rng = np.random.default_rng(11)
    shape = [21, 200, 176]
    coords = _t(random_sparse_coords(rng, 40000, 4, shape), dev)
    outs = []
    for _ in range(2):
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        g = torch.Generator(device=dev).manual_seed(5)
        x = torch.randn(rb.n_out, 64, device=dev, generator=g).requires_grad_(True)
        w = (torch.randn(27, 64, 64, device=dev, generator=g) / 8).requires_grad_(True)
        dy = torch.randn(rb.n_out, 64, device=dev, generator=g)
        y = sparse.sparse_conv(x, w, rb)
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_dense_scatter_and_backward(dev):
    from crbhip import sparse
    rng = np.random.default_rng(5)
    shape = [2, 200, 176]
    coords = random_sparse_coords(rng, 9000, 3, shape)
    order = np.lexsort((coords[:, 3], coords[:, 2], coords[:, 1], coords[:, 0]))
    coords = np.ascontiguousarray(coords[order])
    F_ = rng.normal(size=(len(coords), 128)).astype(np.float32)
    f = _t(F_, dev).requires_grad_(True)
    d = sparse.to_dense(f, _t(coords, dev), 3, shape)
    np.testing.assert_array_equal(d.detach().cpu().numpy(), oracle.dense(F_, coords, 3, shape))
    g = torch.randn_like(d)
    d.backward(g)
    gn = g.cpu().numpy()
    ref = gn[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]]
    np.testing.assert_array_equal(f.grad.cpu().numpy(), ref)


def test_empty_and_tiny_inputs(dev):
    from crbhip import sparse
    shape = [5, 8, 8]
    for n in (0, 1, 3):
        rng = np.random.default_rng(n)
        coords = random_sparse_coords(rng, n, 1, shape, clustered=False) if n else np.zeros((0, 4), np.int32)
        rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
        x = torch.randn(len(coords), 16, device=dev, requires_grad=True)
        w = torch.randn(27, 16, 16, device=dev, requires_grad=True)
        y = sparse.sparse_conv(x, w, rb)
        y.sum().backward()
        assert y.shape == (len(coords), 16)
        if len(coords):
            nbr = oracle.subm_nbr(coords, shape, [3, 3, 3])
            _close(y.detach().cpu().numpy(), oracle.conv_fwd(x.detach().cpu().numpy(), w.detach().cpu().numpy(), nbr))
        else:
            assert float(w.grad.abs().sum()) == 0.0


def test_spconv_module_api_backbone_chain(dev):
    """the layer chain of VoxelBackBone8x (pcdet/models/backbones_3d/spconv_backbone.py:77-117) on one real frame batch:
    shapes, indice_key reuse, and parity of every level against the oracle applied to the same weights"""
    import spconv.pytorch as spconv
    from crbhip import voxel
    torch.manual_seed(0)
    pts, off, _ = kitti_batch(0, 2)
    r = voxel.voxelize(_t(pts, dev), _t(off, dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5, want_voxels=False, want_mean=True)
    x = spconv.SparseConvTensor(r['mean'], r['coords'], [41, 1600, 1408], 2)
    layers = [
        spconv.SubMConv3d(4, 16, 3, padding=1, bias=False, indice_key='subm1'),
        spconv.SubMConv3d(16, 16, 3, padding=1, bias=False, indice_key='subm1'),
        spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='spconv2'),
        spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key='subm2'),
        spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False, indice_key='spconv3'),
        spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key='subm3'),
        spconv.SparseConv3d(64, 64, 3, stride=2, padding=(0, 1, 1), bias=False, indice_key='spconv4'),
        spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key='subm4'),
        spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=0, bias=False, indice_key='spconv_down2'),
    ]
    expect_shapes = [[41, 1600, 1408]] * 2 + [[21, 800, 704]] * 2 + [[11, 400, 352]] * 2 + [[5, 200, 176]] * 2 + \
        [[2, 200, 176]]
    coords = r['coords'].cpu().numpy()
    feats = r['mean'].cpu().numpy()
    shape = [41, 1600, 1408]
    for layer, es in zip(layers, expect_shapes):
        layer = layer.to(dev)
        x = layer(x)
        assert x.spatial_shape == es
        W = layer.weight_kio().detach().cpu().numpy()
        if layer.subm:
            nbr = oracle.subm_nbr(coords, shape, layer.kernel_size)
        else:
            oc, shape = oracle.spconv_out(coords, shape, layer.kernel_size, layer.stride, layer.padding)
            nbr = oracle.spconv_nbr(coords, [s for s in x.indice_dict[layer.indice_key].in_shape], oc,
                                    layer.kernel_size, layer.stride, layer.padding)
            coords = oc
            np.testing.assert_array_equal(x.indices.cpu().numpy(), oc)
        feats = oracle.conv_fwd(feats, W, nbr)
        _close(x.features.detach().cpu().numpy(), feats, rtol=2e-4)
        # keep both chains identical going forward (ReLU keeps magnitudes sane)
        feats = np.maximum(feats, 0)
        x = x.replace_feature(torch.relu(x.features))
    assert set(x.indice_dict.keys()) == {'subm1', 'spconv2', 'subm2', 'spconv3', 'subm3', 'spconv4', 'subm4',
                                         'spconv_down2'}
    d = x.dense()
    assert d.shape == (2, 128, 2, 200, 176)


def _rank_bits_key(mask, chunk, mode):
    """numpy restatement of the sort key of mask_sort_chunks_kernel: 0 = the mask itself; 1 = bits re-ranked rarest offset of
    a 3x3x3 kernel first (corners, edges, faces, centre); 2 = bits re-ranked by their frequency inside the chunk (rarest = most
    significant, ties by bit index)"""
    m = mask.astype(np.int64) & 0xffffffff
    n = len(m)
    out = np.zeros(n, np.int64)
    for c0 in range(0, n, chunk):
        mc = m[c0:c0 + chunk]
        if mode == 0:
            pos = np.arange(32)
        elif mode == 1:
            nxt, pos = [0, 1, 7, 19], np.arange(32)
            for o in range(27):
                nz = (o // 9 != 1) + ((o // 3) % 3 != 1) + (o % 3 != 1)
                pos[o] = nxt[nz]
                nxt[nz] += 1
        else:
            hist = np.array([int(((mc >> b) & 1).sum()) for b in range(32)])
            pos = np.array([int(((hist > hist[b]) | ((hist == hist[b]) & (np.arange(32) < b))).sum()) for b in range(32)])
        r = np.zeros(len(mc), np.int64)
        for b in range(32):
            r |= ((mc >> b) & 1) << int(pos[b])
        out[c0:c0 + chunk] = r
    return out


def test_chunk_mask_sort_is_a_stable_sort_per_chunk(dev):
    """the LDS bitonic chunk sort equals a stable sort by (chunk, key DESCENDING: heaviest tiles first), key = the mask with
    its bits re-ranked rarest first by frequency inside the chunk (_rank_bits_key mode 2; modes 0 / 1 are A/B alternatives of
    the measurement build) — the gather-GEMM result does not depend on it, the MFMA tile fill and the launch tail do"""
    from crbhip import lib, check, ptr, cur_stream
    rng = np.random.default_rng(3)
    mode = 2
    if True:
        for n in (1, 100, 4096, 4097, 50000):
            mask = (rng.integers(0, 1 << 27, n) & rng.integers(0, 1 << 27, n)).astype(np.int32)   # bits at 25 %: skewed by hand below
            mask |= (rng.random(n) < 0.9).astype(np.int32) << 13
            mask[rng.random(n) < 0.5] = 7                         # many ties
            m = _t(mask, dev)
            perm = torch.empty(n, dtype=torch.int32, device=dev)
            check(lib.crb_mask_sort_chunks(ptr(m), n, ptr(perm), cur_stream(dev)), 'sort')
            chunk = lib.crb_mask_sort_chunk_rows()
            key = (np.arange(n) // chunk).astype(np.int64) * (1 << 32) + ((~_rank_bits_key(mask, chunk, mode)) & 0xffffffff)
            np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(key, kind='stable'))


@pytest.mark.parametrize('chunk', [4096, 16384])
def test_radix_chunk_sort_equals_the_lds_sort_and_the_restatement(dev, chunk):
    """crb_mask_sort_rows (ranked keys per chunk + one stable device radix sort) gives the order of crb_mask_sort_chunks for
    4,096-row chunks and the stable (chunk, ranked key descending) order for chunks beyond one workgroup's LDS"""
    from crbhip import lib, check, ptr, cur_stream
    rng = np.random.default_rng(5)
    for n in (1, 5000, 16385, 70001):
        mask = (rng.integers(0, 1 << 27, n) & rng.integers(0, 1 << 27, n)).astype(np.int32)
        mask[rng.random(n) < 0.4] = 21
        m = _t(mask, dev)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        wsb = lib.crb_mask_sort_rows_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        check(lib.crb_mask_sort_rows(ptr(m), n, chunk, ptr(perm), ptr(ws), wsb, cur_stream(dev)), 'rows')
        key = (np.arange(n) // chunk).astype(np.int64) * (1 << 32) + ((~_rank_bits_key(mask, chunk, 2)) & 0xffffffff)
        np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(key, kind='stable'))
        if chunk == lib.crb_mask_sort_chunk_rows():
            perm2 = torch.empty(n, dtype=torch.int32, device=dev)
            check(lib.crb_mask_sort_chunks(ptr(m), n, ptr(perm2), cur_stream(dev)), 'chunks')
            assert torch.equal(perm, perm2)


def test_tile_lpt_perm_orders_whole_tiles_heaviest_first(dev):
    """crb_tile_lpt_perm: output = input perm with its full 64-row tiles moved as units INSIDE their range (8 contiguous
    ranges of ceil(ceil(n/64)/8) tiles), tile weights (popcount of the OR of the tile's row masks) non-increasing within a
    range, trailing partial tile untouched"""
    from crbhip import lib, check, ptr, cur_stream
    rng = np.random.default_rng(5)
    R = lib.crb_tile_lpt_ranges()
    for n in (64, 130, 4096 + 17, 30000):
        mask = (rng.integers(0, 1 << 27, n) & rng.integers(0, 1 << 27, n) & rng.integers(0, 1 << 27, n)).astype(np.int32)
        mask[rng.random(n) < 0.6] = 0                         # spread the tile weights
        perm = np.argsort(mask, kind='stable').astype(np.int32)
        out = torch.empty(n, dtype=torch.int32, device=dev)
        wsb = lib.crb_tile_lpt_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        m_d, p_d = _t(mask, dev), _t(perm, dev)               # named: a temporary would be freed (and reused) after ptr()
        check(lib.crb_tile_lpt_perm(ptr(m_d), ptr(p_d), n, ptr(out), ptr(ws), wsb, cur_stream(dev)), 'lpt')
        o = out.cpu().numpy()
        T = n // 64
        per = -(-(-(-n // 64)) // R)
        np.testing.assert_array_equal(o[T * 64:], perm[T * 64:])

        def weight(rows):
            return bin(int(np.bitwise_or.reduce(mask[rows].astype(np.int64) & 0xffffffff))).count('1')
        for r in range(R):
            t0, t1 = min(r * per, T), min((r + 1) * per, T)
            tin = sorted(tuple(perm[64 * t:64 * t + 64]) for t in range(t0, t1))
            tout = [tuple(o[64 * t:64 * t + 64]) for t in range(t0, t1)]
            assert sorted(tout) == tin, (n, r)
            w = [weight(np.array(t)) for t in tout]
            assert all(a >= b for a, b in zip(w, w[1:])), (n, r, w)


def test_bev_channels_last_scatter_equals_dense_view(dev):
    from crbhip import sparse
    rng = np.random.default_rng(6)
    shape = [2, 200, 176]
    coords = random_sparse_coords(rng, 7000, 3, shape)
    F_ = rng.normal(size=(len(coords), 128)).astype(np.float32)
    f1 = _t(F_, dev).requires_grad_(True)
    f2 = _t(F_, dev).requires_grad_(True)
    a = sparse.to_bev_channels_last(f1, _t(coords, dev), 3, shape)
    b = sparse.to_dense(f2, _t(coords, dev), 3, shape).view(3, 256, 200, 176)
    assert a.is_contiguous(memory_format=torch.channels_last) and a.shape == b.shape
    assert torch.equal(a, b)
    g = torch.randn_like(b)
    a.backward(g)
    b.backward(g)
    assert torch.equal(f1.grad, f2.grad)


def test_fused_bn_relu_max_concat_matches_separate_ops(dev):
    """BatchNorm+ReLU -> max over groups of ns rows -> concat of two scales as one op (crb_bn_relu_max_*) vs the fused
    BN+ReLU op followed by torch's view/max/cat: identical values (same expression per element), gradients w.r.t. the rows
    and the affine parameters (1e-5 of their scale; duplicated rows inside a group tie and may route the gradient to another
    copy in torch, so the rows are distinct here), identical running statistics."""
    from crbhip import bnrelu
    torch.manual_seed(3)
    M = 3001
    specs = [(32, 16), (64, 5)]
    xs = [torch.randn(M * ns, C, device=dev) * 2 + 0.3 for C, ns in specs]
    xs[0][:16 * 7] = -50.0                                   # groups whose every ReLU output is 0
    bns_a = [torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.1).to(dev) for C, _ in specs]
    bns_b = [torch.nn.BatchNorm1d(C, eps=1e-5, momentum=0.1).to(dev) for C, _ in specs]
    with torch.no_grad():
        for a, b in zip(bns_a, bns_b):
            a.weight.uniform_(-1.0, 1.5)
            a.bias.uniform_(-0.5, 0.5)
            b.load_state_dict(a.state_dict())
    xa = [x.clone().requires_grad_(True) for x in xs]
    xb = [x.clone().requires_grad_(True) for x in xs]
    out_a = bnrelu.bn_relu_max_concat(xa, [ns for _, ns in specs], bns_a)
    out_b = torch.cat([bnrelu.bn_relu(x, bn).view(M, ns, C).max(dim=1).values
                       for x, bn, (C, ns) in zip(xb, bns_b, specs)], dim=1)
    assert out_a.shape == (M, 96)
    assert torch.equal(out_a, out_b)
    g = torch.randn_like(out_a)
    out_a.backward(g)
    out_b.backward(g)
    for a, b in zip(xa, xb):
        assert float((a.grad - b.grad).abs().max()) < 1e-5 * max(1.0, float(b.grad.abs().max()))
    for a, b in zip(bns_a, bns_b):
        for pa, pb in ((a.weight, b.weight), (a.bias, b.bias)):
            assert float((pa.grad - pb.grad).abs().max()) < 1e-5 * max(1.0, float(pb.grad.abs().max()))
        assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)
        assert int(a.num_batches_tracked) == 1


@pytest.mark.parametrize('C,n', [(16, 50000), (32, 4097), (64, 2), (128, 30000)])
def test_fused_bn_relu_matches_torch(dev, C, n):
    """fused BatchNorm1d+ReLU (crb_bn_relu_*) vs nn.BatchNorm1d(eps=1e-3, momentum=0.01) + nn.ReLU: outputs, input /
    affine grads and the running statistics; rtol 1e-4 (two-pass float vs torch's Welford)"""
    from crbhip import bnrelu
    torch.manual_seed(C + n)
    x = (torch.randn(n, C, device=dev) * 3 + 1.5)
    dz = torch.randn(n, C, device=dev)
    bn_a = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev)
    bn_b = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev)
    with torch.no_grad():
        bn_a.weight.uniform_(0.5, 1.5)
        bn_a.bias.uniform_(-0.5, 0.5)
        bn_b.load_state_dict(bn_a.state_dict())
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    za = bnrelu.bn_relu(xa, bn_a, relu=True)
    zb = torch.relu(bn_b(xb))
    torch.testing.assert_close(za, zb, rtol=1e-4, atol=1e-4)
    za.backward(dz)
    zb.backward(dz)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(bn_a.weight.grad, bn_b.weight.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(bn_a.bias.grad, bn_b.bias.grad, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(bn_a.running_mean, bn_b.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn_a.running_var, bn_b.running_var, rtol=1e-4, atol=1e-5)
    assert int(bn_a.num_batches_tracked) == 1
    bn_a.eval(); bn_b.eval()
    with torch.no_grad():
        torch.testing.assert_close(bnrelu.bn_relu(x, bn_a), torch.relu(bn_b(x)), rtol=1e-5, atol=1e-5)


def test_sparse_basic_block_with_bias_matches_oracle(dev):
    """SparseBasicBlock of VoxelResBackBone8x (spconv_backbone.py:30-66): two bias=True SubM convs sharing one rulebook,
    BN (train statistics) + ReLU, residual add — forward and every gradient against the oracle convolution with the same
    weights (bias added / reduced on the host in float64)"""
    from functools import partial
    import spconv.pytorch as spconv
    from pcdet.models.backbones_3d.spconv_backbone import SparseBasicBlock
    rng = np.random.default_rng(77)
    torch.manual_seed(3)
    shape = [21, 100, 88]
    coords = random_sparse_coords(rng, 4000, 2, shape)
    n, C = len(coords), 32
    X = rng.normal(size=(n, C)).astype(np.float32)
    dY = rng.normal(size=(n, C)).astype(np.float32)
    blk = SparseBasicBlock(C, C, norm_fn=partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01), indice_key='res1').to(dev)
    blk.train()
    assert blk.conv1.bias is not None and blk.conv2.bias is not None
    with torch.no_grad():
        blk.conv1.bias.uniform_(-0.5, 0.5)
        blk.conv2.bias.uniform_(-0.5, 0.5)
    x = _t(X, dev).requires_grad_(True)
    out = blk(spconv.SparseConvTensor(x, _t(coords, dev), shape, 2))
    out.features.backward(_t(dY, dev))
    # ---- oracle chain in float64 on the host (autograd over the oracle conv)
    nbr = oracle.subm_nbr(coords, shape, [3, 3, 3])

    class OConv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, xx, ww):
            ctx.save_for_backward(xx, ww)
            return torch.from_numpy(oracle.conv_fwd(xx.numpy().astype(np.float32), ww.numpy().astype(np.float32), nbr)).double()

        @staticmethod
        def backward(ctx, g):
            xx, ww = ctx.saved_tensors
            gn = g.numpy().astype(np.float32)
            return (torch.from_numpy(oracle.conv_dgrad(gn, ww.numpy().astype(np.float32), nbr, n)).double(),
                    torch.from_numpy(oracle.conv_wgrad(xx.numpy().astype(np.float32), gn, nbr, 27)).double())
    P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in blk.named_parameters()}
    w1 = blk.conv1.weight_kio().detach().cpu().double().requires_grad_(True)
    w2 = blk.conv2.weight_kio().detach().cpu().double().requires_grad_(True)
    xr = torch.from_numpy(X).double().requires_grad_(True)
    bn = lambda t, k: torch.nn.functional.batch_norm(t, None, None, P[k + '.weight'], P[k + '.bias'], True, 0.0, 1e-3)
    h = torch.relu(bn(OConv.apply(xr, w1) + P['conv1.bias'], 'bn1'))
    h = bn(OConv.apply(h, w2) + P['conv2.bias'], 'bn2')
    ref = torch.relu(h + xr)
    ref.backward(torch.from_numpy(dY).double())
    _close(out.features.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-4)
    _close(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-3)
    _close(_kio_grad(blk.conv1), w1.grad.numpy(), rtol=1e-3)
    _close(_kio_grad(blk.conv2), w2.grad.numpy(), rtol=1e-3)
    named = dict(blk.named_parameters())
    for k in ('bn1.weight', 'bn1.bias', 'bn2.weight', 'bn2.bias'):
        _close(named[k].grad.cpu().numpy(), P[k].grad.numpy(), rtol=1e-3)
    # a bias in front of a train-mode BatchNorm has a mathematically zero gradient (the batch mean is subtracted again):
    # both chains must produce rounding noise only, far below the BN parameter gradients
    scale = float(P['bn2.bias'].grad.abs().max())
    for k in ('conv1.bias', 'conv2.bias'):
        assert float(P[k].grad.abs().max()) < 1e-6 * scale
        assert float(named[k].grad.abs().max()) < 1e-4 * scale, (k, float(named[k].grad.abs().max()), scale)
    # ... so the bias is pinned through the forward instead: eval-mode BN (running statistics) keeps it visible
    blk.eval()
    with torch.no_grad():
        out_e = blk(spconv.SparseConvTensor(_t(X, dev), _t(coords, dev), shape, 2)).features.cpu().numpy()
        st = {k: v.detach().cpu().double() for k, v in blk.state_dict().items()}
        bne = lambda t, k: torch.nn.functional.batch_norm(t, st[k + '.running_mean'], st[k + '.running_var'], st[k + '.weight'],
                                                          st[k + '.bias'], False, 0.0, 1e-3)
        xe = torch.from_numpy(X).double()
        he = torch.relu(bne(OConv.apply(xe, w1.detach()) + st['conv1.bias'], 'bn1'))
        he = bne(OConv.apply(he, w2.detach()) + st['conv2.bias'], 'bn2')
        ref_e = torch.relu(he + xe).numpy()
    _close(out_e, ref_e, rtol=2e-4)
    no_bias = torch.relu(bne(OConv.apply(torch.relu(bne(OConv.apply(xe, w1.detach()), 'bn1')), w2.detach()), 'bn2') + xe).numpy()
    assert np.abs(no_bias - ref_e).max() > 1e-2              # the check would notice a dropped bias


def _kio_grad(conv):
    """gradient of the module's (Cout,k,k,k,Cin) weight viewed in the kernel's (K,Cin,Cout) layout"""
    g = conv.weight.grad
    return g.reshape(g.shape[0], -1, g.shape[-1]).permute(1, 2, 0).contiguous().cpu().numpy()


@pytest.mark.parametrize('kind', ['subm', 'strided'])
def test_compact_table_is_the_sorted_table_and_gives_identical_results(dev, kind):
    """the tables crb_tables_finish leaves for the gather-GEMM: mask + packed present indices decode to exactly the (n,K)
    table (itself bit-exact vs the oracle) in kernel order; the kernel order is the stable per-chunk sort by ranked mask; the
    tile order is the stable heaviest-first order inside each of the 8 XCD ranges; crb_sparse_conv_forward_compact (with and
    without the tile order) == crb_sparse_conv_forward on the (n,K) rows, bit for bit, forward and dgrad shapes"""
    from crbhip import lib, sparse
    rng = np.random.default_rng(31)
    shape = [21, 120, 100]
    coords = random_sparse_coords(rng, 9000, 2, shape)
    with torch.enable_grad():                                  # want_grad: the transposed table is part of the plan
        if kind == 'subm':
            rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
            which = ['nbr']
        else:
            rb = sparse.spconv_rulebook(_t(coords, dev), shape, 2, [3, 3, 3], [2, 2, 2], [1, 1, 1])
            which = ['nbr', 'nbr_t']
    chunk = lib.crb_table_chunk_rows()
    assert chunk == lib.crb_mask_sort_chunk_rows()
    for w_ in which:
        nat = rb.nbr if w_ == 'nbr' else rb.nbr_t
        ct = rb.compact_table(w_)
        n = nat.shape[0]
        assert torch.equal(ct.to_nbr(), nat[ct.perm.long()])
        P = int((nat >= 0).sum())
        assert ct.num_pairs() == P and int(ct.cbase[0]) == 0
        mask = (((nat >= 0).long() << torch.arange(27, device=dev)[None, :]).sum(1)).cpu().numpy()
        key = (np.arange(n) // chunk).astype(np.int64) * (1 << 32) + ((~_rank_bits_key(mask, chunk, 2)) & 0xffffffff)
        np.testing.assert_array_equal(ct.perm.cpu().numpy(), np.argsort(key, kind='stable'))
        # tile order: stable sort by weight (offsets present in any row of the tile), descending, inside each range
        sm = mask[ct.perm.cpu().numpy()]
        tiles_all, full = (n + 63) // 64, n // 64
        wt = np.array([bin(int(np.bitwise_or.reduce(sm[t * 64:(t + 1) * 64]))).count('1') for t in range(full)])
        per = (tiles_all + 7) // 8
        expect = np.arange(tiles_all)
        for r in range(8):
            lo, hi = r * per, min((r + 1) * per, full)
            if lo < hi:
                expect[lo:hi] = lo + np.argsort(-wt[lo:hi], kind='stable')
        np.testing.assert_array_equal(ct.order.cpu().numpy(), expect)
        full_rows, perm = rb.sorted_table(w_)                          # rows physically in dispatch order (v1 kernels)
        assert torch.equal(full_rows, nat[perm.long()]) and sorted(perm.cpu().tolist()) == list(range(n))
        n_in = rb.n_in if w_ == 'nbr' else rb.n_out
        ct_plain = sparse.CompactTable(ct.cmask, ct.cbase, ct.packed, ct.perm, ct.n, ct.K, None)
        for cin, cout in ((4, 16), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 64), (64, 128)):
            assert lib.crb_sparse_conv_compact_supported(cin, cout) == 1
            x = torch.randn(n_in, cin, device=dev)
            w = torch.randn(27, cin, cout, device=dev) / 8
            a = sparse._conv_forward_raw(x, w, (full_rows, perm), n)
            b = sparse._conv_forward_raw(x, w, ct, n)
            c = sparse._conv_forward_raw(x, w, ct_plain, n)
            assert torch.equal(a, b) and torch.equal(a, c), (cin, cout, float((a - b).abs().max()))
    assert lib.crb_sparse_conv_compact_supported(5, 16) == 0 and lib.crb_sparse_conv_compact_supported(128, 64) == 0
    # natural row order (no permutation) and an all-empty table
    ct0 = sparse._compact(rb.nbr, None, 27)
    assert torch.equal(ct0.to_nbr(), rb.nbr)
    empty = torch.full((130, 27), -1, dtype=torch.int32, device=dev)
    cte = sparse._compact(empty, None, 27)
    assert cte.num_pairs() == 0
    y = sparse._conv_forward_raw(torch.randn(5, 16, device=dev), torch.randn(27, 16, 16, device=dev), cte, 130)
    assert float(y.abs().max()) == 0.0
    rbe = sparse.Rulebook(empty, None, 130, 130, [3, 3, 3], [1, 1, 1], [1, 1, 1], True, shape, shape, None)
    cte2 = rbe.compact_table('nbr')                                     # the fused path on an all-empty table
    assert cte2.num_pairs() == 0 and torch.equal(cte2.to_nbr(), empty[cte2.perm.long()])
    assert int(rbe.pairs()[2][-1]) == 0


def test_planned_chain_equals_layer_by_layer_rulebooks_and_the_oracle(dev):
    """crbhip.sparse.build_rulebooks on the whole geometry chain of VoxelBackBone8x (subm, k3 s2 p1, subm, k3 s2 p1, subm,
    k3 s2 p(0,1,1), subm, k(3,1,1) s(2,1,1) p0): every level marked from the previous level's BITMAP, one host read-back,
    SubM tables of the chain's own levels through the rank table instead of a hash == the rulebooks built one layer at a
    time == the oracle (output sets, both tables of the strided convs, SubM tables, pair lists)"""
    from crbhip import sparse
    rng = np.random.default_rng(5)
    shape = [41, 160, 144]
    coords = random_sparse_coords(rng, 12000, 3, shape)
    geoms = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
             ((3, 1, 1), (2, 1, 1), (0, 0, 0))]
    specs = []
    for g in geoms:
        specs += [('subm', (3, 3, 3)), ('spconv',) + g]
    c = _t(coords, dev)
    with torch.enable_grad():
        books = sparse.build_rulebooks(c, shape, 3, specs, want_grad=True)
    cur, cur_shape = c, shape
    oc_np, oshape_np = coords, shape
    for k, (ks, st, pd) in enumerate(geoms):
        sub, b = books[2 * k], books[2 * k + 1]
        np.testing.assert_array_equal(sub.nbr.cpu().numpy(), oracle.subm_nbr(oc_np, oshape_np, [3, 3, 3]))
        assert torch.equal(sparse.subm_rulebook(cur, cur_shape, [3, 3, 3]).nbr, sub.nbr)         # hash path == rank path
        a = sparse.spconv_rulebook(cur, cur_shape, 3, ks, st, pd)
        assert a.n_out == b.n_out and a.out_shape == b.out_shape
        assert torch.equal(a.out_coords, b.out_coords) and torch.equal(a.nbr, b.nbr) and torch.equal(a.nbr_t, b.nbr_t)
        nxt, nshape = oracle.spconv_out(oc_np, oshape_np, ks, st, pd)
        np.testing.assert_array_equal(b.out_coords.cpu().numpy(), nxt)
        nbr_ref = oracle.spconv_nbr(oc_np, oshape_np, nxt, ks, st, pd)
        np.testing.assert_array_equal(b.nbr.cpu().numpy(), nbr_ref)
        i, o = np.nonzero(nbr_ref >= 0)
        ref_t = np.full((len(oc_np), nbr_ref.shape[1]), -1, np.int32)
        ref_t[nbr_ref[i, o], o] = i
        np.testing.assert_array_equal(b.nbr_t.cpu().numpy(), ref_t)
        pin, pout, pstart = [x.cpu().numpy() for x in b.pairs()[:3]]
        K = nbr_ref.shape[1]
        assert int(pstart[K]) == (nbr_ref >= 0).sum()
        for o_ in range(K):
            rows = np.nonzero(nbr_ref[:, o_] >= 0)[0]
            np.testing.assert_array_equal(pout[pstart[o_]:pstart[o_ + 1]], rows)
            np.testing.assert_array_equal(pin[pstart[o_]:pstart[o_ + 1]], nbr_ref[rows, o_])
        oc_np, oshape_np = nxt, nshape
        cur, cur_shape = b.out_coords.contiguous(), b.out_shape


def test_backbone_with_planned_indices_equals_unplanned(dev):
    from pcdet.models.backbones_3d import spconv_backbone as sb
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(second_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.eval()
    pts, off, gt = kitti_batch(0, 2)
    bidx = np.repeat(np.arange(2, dtype=np.float32), np.diff(off))[:, None]
    outs = []
    for plan in (True, False):
        sb.PLAN_INDICES = plan
        bd = {'points': _t(np.concatenate([bidx, pts], 1), dev), 'point_frame_offsets': _t(off, dev), 'batch_size': 2}
        with torch.no_grad():
            bd = model.vfe(bd)
            bd = model.backbone_3d(bd)
        outs.append(bd['encoded_spconv_tensor'])
    sb.PLAN_INDICES = True
    assert torch.equal(outs[0].indices, outs[1].indices) and torch.equal(outs[0].features, outs[1].features)
    assert set(outs[0].indice_dict.keys()) == set(outs[1].indice_dict.keys())


def test_windowed_wgrad_equals_default_wgrad(dev):
    """crb_sparse_conv_wgrad_windowed (work cut by windows of output rows, opt-in) == the default wgrad == the oracle"""
    from crbhip import sparse
    rng = np.random.default_rng(91)
    shape = [21, 160, 140]
    coords = random_sparse_coords(rng, 30000, 3, shape)
    rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
    n = rb.n_out
    nbr = oracle.subm_nbr(coords, shape, [3, 3, 3])
    for cin, cout in ((32, 32), (32, 64), (64, 64)):
        X = rng.normal(size=(n, cin)).astype(np.float32)
        dY = rng.normal(size=(n, cout)).astype(np.float32)
        ref = oracle.conv_wgrad(X, dY, nbr, 27)
        got = {}
        for w_ in (False, True):
            sparse.WGRAD_WINDOWED = w_
            got[w_] = sparse._conv_wgrad_raw(_t(X, dev), _t(dY, dev), rb.pairs(), 27).cpu().numpy()
        sparse.WGRAD_WINDOWED = False
        _close(got[False], ref)
        _close(got[True], ref)
    bnd = rb.pairs()[3].cpu().numpy()
    pstart = rb.pairs()[2].cpu().numpy()
    assert (bnd[:, 0] == pstart[:-1]).all() and (bnd[:, -1] == pstart[1:]).all() and (np.diff(bnd, axis=1) >= 0).all()


BF16X3_BOUND = 2.0 ** -16          # the stated contract of crb_sparse_conv_forward_bf16x3 (include/crb_hip.h)
F32_ACC_SLACK = 2.0 ** -20         # f32 accumulation of <= 27*128 terms on top (either kernel has it)


def _bf16x3_check(got, exact, absum):
    err = np.abs(got.astype(np.float64) - exact)
    lim = (BF16X3_BOUND + F32_ACC_SLACK) * absum
    assert (err <= lim).all(), (float((err / np.maximum(absum, 1e-30)).max()), BF16X3_BOUND)
    return float((err / np.maximum(absum, 1e-30)).max())


@pytest.mark.parametrize('cin,cout', [(32, 32), (32, 64), (64, 64), (64, 32)])
def test_bf16x3_subm_forward_and_dgrad_within_the_stated_bound(dev, cin, cout):
    """OPT-IN arithmetic (sparse_conv(..., arithmetic='bf16x3')): operands split into two bf16 values, 3 bf16 MFMA passes.
    Stated contract: |y - y_exact| <= 2^-16 * sum |x||w| over the products of the output element; checked against the
    double-accumulating oracle on rows whose magnitudes span six decades; dgrad runs the same kernel on the transposed
    table (here also the (64,32)/(32,64) pairs); wgrad keeps exact f32. The default path must stay exact f32."""
    from conftest import require_measure_lib
    require_measure_lib()
    from crbhip import sparse
    rng = np.random.default_rng(300 + cin + cout)
    shape = [21, 100, 88]
    coords = random_sparse_coords(rng, 5000, 2, shape)
    n = len(coords)
    X = (rng.normal(size=(n, cin)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    W = (rng.normal(size=(27, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    dY = (rng.normal(size=(n, cout)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))).astype(np.float32)
    nbr = oracle.subm_nbr(coords, shape, [3, 3, 3])
    rb = sparse.subm_rulebook(_t(coords, dev), shape, [3, 3, 3])
    x = _t(X, dev).requires_grad_(True)
    w = _t(W, dev).requires_grad_(True)
    y_f32 = sparse.sparse_conv(x, w, rb).detach()
    y = sparse.sparse_conv(x, w, rb, False, 'bf16x3')
    y.backward(_t(dY, dev))
    assert torch.equal(sparse.sparse_conv(x, w, rb).detach(), y_f32)       # the opt-in call left no state behind
    worst = _bf16x3_check(y.detach().cpu().numpy(), oracle.conv_fwd(X, W, nbr).astype(np.float64),
                          oracle.conv_fwd(np.abs(X), np.abs(W), nbr).astype(np.float64))
    _bf16x3_check(x.grad.cpu().numpy(), oracle.conv_dgrad(dY, W, nbr, n).astype(np.float64),
                  oracle.conv_dgrad(np.abs(dY), np.abs(W), nbr, n).astype(np.float64))
    _close(w.grad.cpu().numpy(), oracle.conv_wgrad(X, dY, nbr, 27), rtol=2e-4)          # exact-f32 wgrad, unchanged
    assert not torch.equal(y.detach(), y_f32) and worst > 2.0 ** -24       # the opt-in kernel really ran
    _close(y_f32.cpu().numpy(), oracle.conv_fwd(X, W, nbr))


@pytest.mark.parametrize('cin,cout,ks,st,pd', [(64, 128, (3, 1, 1), (2, 1, 1), (0, 0, 0)),
                                                (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                                (64, 64, (3, 3, 3), (2, 2, 2), (0, 1, 1))])
def test_bf16x3_strided_conv_out_layer(dev, cin, cout, ks, st, pd):
    """strided layers under the opt-in contract: the (3,1,1)/(2,1,1) 64->128 conv_out geometry with its 128->64 dgrad, and the
    3x3x3 stride-2 down-sampling convs (32->64, 64->64) with their dgrads through the transposed table"""
    from conftest import require_measure_lib
    require_measure_lib()
    from crbhip import sparse
    rng = np.random.default_rng(77 + cin + cout)
    shape = [11, 100, 88]
    coords = random_sparse_coords(rng, 6000, 2, shape)
    n = len(coords)
    K = ks[0] * ks[1] * ks[2]
    X = rng.normal(size=(n, cin)).astype(np.float32)
    W = (rng.normal(size=(K, cin, cout)) / 8).astype(np.float32)
    rb = sparse.spconv_rulebook(_t(coords, dev), shape, 2, ks, st, pd)
    dY = rng.normal(size=(rb.n_out, cout)).astype(np.float32)
    x = _t(X, dev).requires_grad_(True)
    w = _t(W, dev).requires_grad_(True)
    y = sparse.sparse_conv(x, w, rb, False, 'bf16x3')
    y.backward(_t(dY, dev))
    oc, _ = oracle.spconv_out(coords, shape, ks, st, pd)
    nbr = oracle.spconv_nbr(coords, shape, oc, ks, st, pd)
    _bf16x3_check(y.detach().cpu().numpy(), oracle.conv_fwd(X, W, nbr).astype(np.float64),
                  oracle.conv_fwd(np.abs(X), np.abs(W), nbr).astype(np.float64))
    _bf16x3_check(x.grad.cpu().numpy(), oracle.conv_dgrad(dY, W, nbr, n).astype(np.float64),
                  oracle.conv_dgrad(np.abs(dY), np.abs(W), nbr, n).astype(np.float64))


@pytest.mark.parametrize('cin,cout,bias', [(16, 16, False), (32, 64, True), (64, 64, False), (64, 128, False)])
def test_eval_conv_bn_relu_epilogue_equals_the_three_modules(dev, cin, cout, bias):
    """inference: SparseSequential(conv, BatchNorm1d, ReLU) with BatchNorm in eval mode runs as ONE launch (normalisation and
    ReLU on the accumulator, crb_sparse_conv_forward_compact_bn). Same arithmetic order as conv -> crb_bn_relu_apply:
    results equal the unfused path to 1e-6 of the largest magnitude (rsqrt of the epilogue vs torch.rsqrt), for a SubM
    and a strided block, with and without conv bias; with grad enabled or BatchNorm in train mode nothing is fused."""
    import spconv.pytorch as spconv
    from spconv.pytorch import modules as spm
    rng = np.random.default_rng(900 + cin + cout)
    shape = [21, 100, 88]
    coords = random_sparse_coords(rng, 5000, 2, shape)
    n = len(coords)
    torch.manual_seed(cin * 131 + cout)
    blocks = [spconv.SparseSequential(spconv.SubMConv3d(cin, cout, 3, bias=bias, indice_key='s'),
                                      torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), torch.nn.ReLU()),
              spconv.SparseSequential(spconv.SparseConv3d(cin, cout, 3, stride=2, padding=1, bias=bias, indice_key='d'),
                                      torch.nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01), torch.nn.ReLU())]
    for blk in blocks:
        blk.to(dev).eval()
        with torch.no_grad():
            blk[1].running_mean.normal_(0, 0.5)
            blk[1].running_var.uniform_(0.3, 2.0)
            blk[1].weight.uniform_(0.5, 1.5)
            blk[1].bias.normal_(0, 0.3)
        x = spconv.SparseConvTensor(_t(rng.normal(size=(n, cin)).astype(np.float32), dev), _t(coords, dev), shape, 2)
        with torch.no_grad():
            fused = blk(x).features
            spm.FUSE_CONV_BN_EVAL = False
            try:
                plain = blk(x).features
            finally:
                spm.FUSE_CONV_BN_EVAL = True
        assert fused.shape == plain.shape and float((plain > 0).float().mean()) > 0.2
        tol = 1e-6 * float(plain.abs().max())
        assert float((fused - plain).abs().max()) <= tol, float((fused - plain).abs().max())
        with torch.enable_grad():                      # grad mode: the module path, untouched
            y = blk(spconv.SparseConvTensor(x.features.clone().requires_grad_(True), x.indices, shape, 2)).features
        assert y.requires_grad and float((y.detach() - plain).abs().max()) <= 1e-5 * float(plain.abs().max())


def test_sparse_inverse_conv3d_matches_the_transposed_rulebook_of_the_oracle(dev):
    """SparseInverseConv3d (spconv/pytorch/conv.py:163-180 in spconv 2.1; the reference's UNet decoder,
    pcdet/models/backbones_3d/spconv_unet.py:100-107): it re-uses the rulebook its indice_key names with input and output
    swapped — fine row i receives sum_o x_coarse[nbr_t[i,o]] W[o], where nbr_t is the exact transpose of the forward
    table, and its output lives on the forward conv's INPUT coordinates. Forward, dgrad and wgrad vs the oracle."""
    import spconv.pytorch as spconv
    rng = np.random.default_rng(77)
    shape = [21, 120, 100]
    coords = random_sparse_coords(rng, 9000, 2, shape)
    n = len(coords)
    torch.manual_seed(5)
    down = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='spconv2').to(dev)
    up = spconv.SparseInverseConv3d(32, 16, 3, indice_key='spconv2', bias=False).to(dev)
    X = rng.normal(size=(n, 16)).astype(np.float32)
    x = spconv.SparseConvTensor(_t(X, dev), _t(coords, dev), shape, 2)
    mid = down(x)
    Z = rng.normal(size=(mid.features.shape[0], 32)).astype(np.float32)        # fresh coarse features: test `up` alone
    z = _t(Z, dev).requires_grad_(True)
    out = up(mid.replace_feature(z))
    assert out.spatial_shape == shape
    np.testing.assert_array_equal(out.indices.cpu().numpy(), coords)            # back on the fine active set, same row order
    dY = rng.normal(size=(n, 16)).astype(np.float32)
    out.features.backward(_t(dY, dev))
    oc, oshape = oracle.spconv_out(coords, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    np.testing.assert_array_equal(mid.indices.cpu().numpy(), oc)
    nbr = oracle.spconv_nbr(coords, shape, oc, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    i, o = np.nonzero(nbr >= 0)
    nbr_t = np.full((n, 27), -1, np.int32)
    nbr_t[nbr[i, o], o] = i
    W = up.weight_kio().detach().cpu().numpy()
    _close(out.features.detach().cpu().numpy(), oracle.conv_fwd(Z, W, nbr_t))
    _close(z.grad.cpu().numpy(), oracle.conv_dgrad(dY, W, nbr_t, len(oc)))
    _close(_kio_grad(up), oracle.conv_wgrad(Z, dY, nbr_t, 27))


@pytest.mark.parametrize('strided', [False, True])
def test_2d_sparse_convs_match_the_oracle_on_a_one_slice_volume(dev, strided):
    """SubMConv2d / SparseConv2d (named in SURVEY §8b): indices (N,3) [b,y,x]; the oracle sees the same sites as a volume
    with D = 1 and a (1,k,k) kernel"""
    import spconv.pytorch as spconv
    rng = np.random.default_rng(78)
    H, W_ = 150, 130
    c3 = random_sparse_coords(rng, 7000, 2, [1, H, W_], clustered=False)
    c2 = np.ascontiguousarray(c3[:, [0, 2, 3]])
    n = len(c2)
    torch.manual_seed(6)
    if strided:
        conv = spconv.SparseConv2d(16, 32, 3, stride=2, padding=1, bias=True, indice_key='sp').to(dev)
    else:
        conv = spconv.SubMConv2d(16, 32, 3, padding=1, bias=True, indice_key='sm').to(dev)
    X = rng.normal(size=(n, 16)).astype(np.float32)
    x = _t(X, dev).requires_grad_(True)
    y = conv(spconv.SparseConvTensor(x, _t(c2, dev), [H, W_], 2))
    if strided:
        oc, oshape = oracle.spconv_out(c3, [1, H, W_], [1, 3, 3], [1, 2, 2], [0, 1, 1])
        nbr = oracle.spconv_nbr(c3, [1, H, W_], oc, [1, 3, 3], [1, 2, 2], [0, 1, 1])
        assert y.spatial_shape == oshape[1:]
        np.testing.assert_array_equal(y.indices.cpu().numpy(), oc[:, [0, 2, 3]])
    else:
        nbr = oracle.subm_nbr(c3, [1, H, W_], [1, 3, 3])
        assert y.spatial_shape == [H, W_]
        np.testing.assert_array_equal(y.indices.cpu().numpy(), c2)
    dY = rng.normal(size=(nbr.shape[0], 32)).astype(np.float32)
    y.features.backward(_t(dY, dev))
    Wk = conv.weight_kio().detach().cpu().numpy()
    _close(y.features.detach().cpu().numpy(), oracle.conv_fwd(X, Wk, nbr) + conv.bias.detach().cpu().numpy())
    _close(x.grad.cpu().numpy(), oracle.conv_dgrad(dY, Wk, nbr, n))
    _close(_kio_grad(conv), oracle.conv_wgrad(X, dY, nbr, 9))
    _close(conv.bias.grad.cpu().numpy(), dY.sum(0), rtol=2e-4)
    d = y.dense()
    assert d.shape == (2, 32) + tuple(y.spatial_shape)


def test_prepared_weight_layouts_equal_the_per_layer_copies(dev):
    """crb_sparse_weights_multi (forward and input-gradient operands of all layers of a step in one launch) against the per-layer
    permute / flip / transpose copies: bit-equal, submanifold and strided layers, kernel sizes 3 and (3,1,1), 40 layers = two launches;
    an updated parameter (new autograd version) is not served from the prepared set"""
    import spconv.pytorch as spconv
    from crbhip import sparse as S
    torch.manual_seed(5)
    convs = [spconv.SubMConv3d(4, 16, 3, bias=False, indice_key='a'), spconv.SubMConv3d(16, 16, 3, bias=False, indice_key='a'),
             spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False, indice_key='b'),
             spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), bias=False, indice_key='c')]
    convs += [spconv.SubMConv3d(32, 32, 3, bias=False, indice_key='d%d' % k) for k in range(36)]
    convs = [c.to(dev) for c in convs]
    S._PREP_W.clear()
    S._PREP_WD.clear()
    plain = []
    for c in convs:
        K = c.weight.numel() // (c.out_channels * c.in_channels)
        kio = c.weight.detach().reshape(c.out_channels, K, c.in_channels).permute(1, 2, 0).contiguous()
        plain.append((kio, (kio.flip(0) if c.subm else kio).transpose(1, 2).contiguous()))
    assert S.prepare_weights(convs) == len(convs)
    for c, (kio, wd) in zip(convs, plain):
        got = c.weight_kio()
        assert got.data_ptr() == S._PREP_W[S._wkey(c.weight)][0].data_ptr() and torch.equal(got, kio)
        assert torch.equal(S._dgrad_weights(got.detach(), c.subm), wd)
        assert S._dgrad_weights(got.detach(), not c.subm).data_ptr() != S._PREP_WD[got.data_ptr()][1].data_ptr()
    # gradient of the layout op: the (K,Cin,Cout) gradient back in the parameter's layout
    g = torch.randn_like(plain[2][0])
    w = convs[2].weight
    w.grad = None
    convs[2].weight_kio().backward(g)
    assert torch.equal(w.grad, g.permute(2, 0, 1).reshape(w.shape))
    with torch.no_grad():
        convs[0].weight.add_(1.0)
    fresh = convs[0].weight_kio()
    assert S._wkey(convs[0].weight) not in S._PREP_W and not torch.equal(fresh, plain[0][0])
    S._PREP_W.clear()
    S._PREP_WD.clear()
