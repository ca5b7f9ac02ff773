"""CPU: pin the sparse-conv oracle (oracle/sparse_conv_oracle.c) against an independent definition:
torch.nn.functional.conv3d on the densified tensor, masked to the analytically derived output active set
(spconv itself is absent, so the dense convolution is the pin)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from synth import random_sparse_coords


def _dense_in(feat, coords, B, shape):
    return torch.from_numpy(oracle.dense(feat, coords, B, shape))


@pytest.mark.parametrize('ks,st,pd', [((3, 3, 3), (1, 1, 1), (1, 1, 1))])
def test_subm_matches_dense_conv(ks, st, pd):
    rng = np.random.default_rng(1)
    B, shape, cin, cout = 2, [9, 14, 12], 5, 7
    coords = random_sparse_coords(rng, 300, B, shape)
    X = rng.normal(size=(len(coords), cin)).astype(np.float32)
    W = rng.normal(size=(27, cin, cout)).astype(np.float32)
    nbr = oracle.subm_nbr(coords, shape, ks)
    Y = oracle.conv_fwd(X, W, nbr)
    # torch weight (cout,cin,kd,kh,kw)
    Wt = torch.from_numpy(W).reshape(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
    dense = F.conv3d(_dense_in(X, coords, B, shape), Wt, padding=1)
    ref = dense[coords[:, 0], :, coords[:, 1], coords[:, 2], coords[:, 3]].numpy()
    np.testing.assert_allclose(Y, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('ks,st,pd', [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)),
                                      ((3, 1, 1), (2, 1, 1), (0, 0, 0))])
def test_strided_matches_dense_conv(ks, st, pd):
    rng = np.random.default_rng(2)
    B, shape, cin, cout = 2, [11, 16, 13], 4, 6
    coords = random_sparse_coords(rng, 400, B, shape)
    X = rng.normal(size=(len(coords), cin)).astype(np.float32)
    K = ks[0] * ks[1] * ks[2]
    W = rng.normal(size=(K, cin, cout)).astype(np.float32)
    out_coords, oshape = oracle.spconv_out(coords, shape, ks, st, pd)
    nbr = oracle.spconv_nbr(coords, shape, out_coords, ks, st, pd)
    Y = oracle.conv_fwd(X, W, nbr)
    Wt = torch.from_numpy(W).reshape(*ks, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
    dense = F.conv3d(_dense_in(X, coords, B, shape), Wt, stride=st, padding=pd)
    assert list(dense.shape[2:]) == oshape
    # active set = receptive field contains >= 1 active input  <=> conv of the occupancy mask with ones > 0
    occ = _dense_in(np.ones((len(coords), 1), np.float32), coords, B, shape)
    cnt = F.conv3d(occ, torch.ones(1, 1, *ks), stride=st, padding=pd)[:, 0]
    act = torch.nonzero(cnt > 0.5).numpy().astype(np.int32)     # ascending (b,z,y,x)
    np.testing.assert_array_equal(out_coords, act)
    ref = dense[act[:, 0], :, act[:, 1], act[:, 2], act[:, 3]].numpy()
    np.testing.assert_allclose(Y, ref, rtol=1e-4, atol=1e-4)
    # every (output, offset) entry appears with multiplicity cnt
    assert (nbr >= 0).sum() == int(cnt.sum().item())


def test_grads_match_autograd_of_dense_conv():
    rng = np.random.default_rng(3)
    B, shape, cin, cout = 1, [7, 9, 8], 3, 4
    ks, st, pd = (3, 3, 3), (2, 2, 2), (1, 1, 1)
    coords = random_sparse_coords(rng, 120, B, shape)
    X = rng.normal(size=(len(coords), cin)).astype(np.float32)
    W = rng.normal(size=(27, cin, cout)).astype(np.float32)
    out_coords, oshape = oracle.spconv_out(coords, shape, ks, st, pd)
    nbr = oracle.spconv_nbr(coords, shape, out_coords, ks, st, pd)
    dY = rng.normal(size=(len(out_coords), cout)).astype(np.float32)
    dX = oracle.conv_dgrad(dY, W, nbr, len(coords))
    dW = oracle.conv_wgrad(X, dY, nbr, 27)
    Xt = torch.from_numpy(X).requires_grad_(True)
    Wk = torch.from_numpy(W).requires_grad_(True)
    c = torch.from_numpy(coords).long()
    dense_in = torch.zeros(B, *shape, cin).index_put((c[:, 0], c[:, 1], c[:, 2], c[:, 3]), Xt)
    dense_in = dense_in.permute(0, 4, 1, 2, 3)
    Wt = Wk.reshape(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2)
    out = F.conv3d(dense_in, Wt, stride=st, padding=pd)
    oc = torch.from_numpy(out_coords).long()
    y = out[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]]
    (y * torch.from_numpy(dY)).sum().backward()
    np.testing.assert_allclose(dX, Xt.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dW, Wk.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_voxelizer_oracle_basic_properties():
    from synth import kitti_frame, KITTI_RANGE, KITTI_VOXEL
    pts, _ = kitti_frame(3)
    v, c, n = oracle.voxelize_frame(pts, KITTI_RANGE[:3], KITTI_VOXEL, [1408, 1600, 40], 16000, 5)
    assert len(np.unique(c, axis=0)) == len(c)
    assert n.min() >= 1 and n.max() <= 5
    # first point of voxel m is the first point (input order) that falls in it; voxels appear in first-point order
    key = (np.floor((pts[:, :3] - np.float32(KITTI_RANGE[:3])) / np.float32(KITTI_VOXEL))).astype(np.int64)
    lin = (key[:, 2] * 1600 + key[:, 1]) * 1408 + key[:, 0]
    _, first = np.unique(lin, return_index=True)
    first = np.sort(first)[:16000]
    np.testing.assert_array_equal(v[:, 0, :], pts[first])
    np.testing.assert_array_equal(c, key[first][:, ::-1])
    m = oracle.mean_vfe(v, n)
    np.testing.assert_allclose(m, v.sum(1) / n[:, None], rtol=1e-6)
