"""Rows a2 / a19 / a20 of SURVEY §8 against goldens produced by the reference's own code
(tests/golden/make_goldens.py:gen_glue): collate_batch, the eval DistributedSampler (-> scoring.shard_indices), the BEV
bilinear lookup of VoxelSetAbstraction and PointHeadSimple targets + loss."""
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_glue.npz'))


def _frames():
    frames = []
    for i in range(3):
        f = {k: G['col_in%d_%s' % (i, k)] for k in ('points', 'voxels', 'voxel_coords', 'voxel_num_points', 'gt_boxes')}
        f['frame_id'] = str(G['col_in_frame_id'][i])
        f['use_lead_xyz'] = True
        frames.append(f)
    return frames


def test_collate_batch_matches_reference_collate():
    """DatasetTemplate.collate_batch (dataset.py:160-229): same keys, dtypes and values; the mirror adds only
    'point_frame_offsets' (the device voxelizer's frame boundaries)"""
    from pcdet.datasets import SyntheticDataset
    got = SyntheticDataset.collate_batch(_frames())
    ref_keys = {k[len('col_out_'):] for k in G.files if k.startswith('col_out_')}
    assert set(got.keys()) - ref_keys == {'point_frame_offsets'} and ref_keys <= set(got.keys())
    for k in ref_keys:
        ref = G['col_out_' + k]
        g = np.asarray(got[k])
        assert g.shape == ref.shape and g.dtype == ref.dtype, (k, g.dtype, ref.dtype, g.shape, ref.shape)
        np.testing.assert_array_equal(g, ref, err_msg=k)
    np.testing.assert_array_equal(got['point_frame_offsets'], [0, 50, 81, 125])


def test_load_data_to_gpu_dtypes_follow_reference_rule():
    """models/__init__.py:23-34: every ndarray becomes float32 except the bookkeeping keys (frame_id stays numpy) —
    checked on the host side of the conversion (no device needed): the table the mirror's loader is driven by"""
    from pcdet.models import _HOST_ONLY_KEYS, _INT_KEYS
    assert 'frame_id' in _HOST_ONLY_KEYS and 'voxel_coords' not in _INT_KEYS and 'voxel_num_points' not in _INT_KEYS


def test_shard_indices_and_sampler_match_reference_sampler_fixture():
    """the reference's own DistributedSampler(shuffle=False) index lists (datasets/__init__.py:26-46) == scoring.shard_indices
    == the mirror sampler, for every rank"""
    from pcdet.datasets.sampler import DistributedSampler
    from pcdet.query_strategies import scoring
    for key in [k for k in G.files if k.startswith('sampler_')]:
        n, world = [int(v) for v in key.split('_')[1:]]
        ref = G[key]
        assert ref.shape[0] == world
        for r in range(world):
            idx, per = scoring.shard_indices(n, r, world)
            assert idx == ref[r].tolist() and per == ref.shape[1]
            assert list(DistributedSampler(list(range(n)), world, r, shuffle=False)) == ref[r].tolist()


def _vsa_stub():
    from pcdet.models.backbones_3d.pfe.voxel_set_abstraction import VoxelSetAbstraction
    return types.SimpleNamespace(voxel_size=[0.05, 0.05, 0.1],
                                 point_cloud_range=np.array([0, -40, -3, 70.4, 40, 1], np.float32)), VoxelSetAbstraction


def _check_bilinear(device, tol=1e-6):
    from pcdet.models.backbones_3d.pfe.voxel_set_abstraction import bilinear_interpolate_torch
    t = lambda a: torch.from_numpy(a).to(device)
    got = bilinear_interpolate_torch(t(G['bil_im']), t(G['bil_x']), t(G['bil_y'])).cpu().numpy()
    np.testing.assert_allclose(got, G['bil_out'], rtol=tol, atol=tol)
    fake, VSA = _vsa_stub()
    got = VSA.interpolate_from_bev_features(fake, t(G['bev_kp']), t(G['bev_map']), 3, 8).cpu().numpy()
    np.testing.assert_allclose(got, G['bev_out'], rtol=tol, atol=tol)


def test_bilinear_bev_lookup_matches_reference_cpu():
    _check_bilinear('cpu')


@pytest.mark.gpu
def test_bilinear_bev_lookup_matches_reference_gpu(dev):
    # the device's elementwise kernels contract a*b+c into FMAs: the four tap weights (products of differences of
    # pixel coordinates up to ~25, ulp 2e-6) and the weighted sum differ from the CPU's by a few 1e-6 on O(1) features
    _check_bilinear(dev, tol=2e-5)


@pytest.mark.gpu
def test_point_head_targets_and_loss_match_reference(dev):
    """PointHeadSimple forward (train) -> assign_stack_targets (HIP points-in-boxes, one batched launch per box set) ->
    focal loss + gradient, against the reference head with the same weights"""
    from pcdet.config import EasyDict
    from pcdet.models.dense_heads.point_head_simple import PointHeadSimple
    cfg = EasyDict({'NAME': 'PointHeadSimple', 'CLS_FC': [16, 16], 'CLASS_AGNOSTIC': True,
                    'USE_POINT_FEATURES_BEFORE_FUSION': True, 'NUM_KEYPOINTS': 64,
                    'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2]},
                    'LOSS_CONFIG': {'LOSS_REG': 'smooth-l1', 'LOSS_WEIGHTS': {'point_cls_weight': 1.0}}})
    head = PointHeadSimple(num_class=1, input_channels=12, model_cfg=cfg)
    state = {k[len('ph_state/'):]: torch.from_numpy(G[k]) for k in G.files if k.startswith('ph_state/')}
    head.load_state_dict(state)
    head = head.to(dev).train()
    feats = torch.from_numpy(G['ph_feats']).to(dev).requires_grad_(True)
    bd = head({'point_features_before_fusion': feats, 'point_features': feats,
               'point_coords': torch.from_numpy(G['ph_coords']).to(dev), 'gt_boxes': torch.from_numpy(G['ph_gt']).to(dev),
               'batch_size': 3})
    np.testing.assert_array_equal(head.forward_ret_dict['point_cls_labels'].cpu().numpy(), G['ph_labels'])
    np.testing.assert_allclose(bd['point_cls_scores'].detach().cpu().numpy(), G['ph_scores'], rtol=1e-5, atol=1e-6)
    loss, tb = head.get_loss()
    loss.backward()
    np.testing.assert_allclose([float(loss), float(tb['point_loss_cls']), float(tb['point_pos_num'])], G['ph_loss'],
                               rtol=1e-5)
    np.testing.assert_allclose(feats.grad.cpu().numpy(), G['ph_feats_grad'], rtol=1e-4, atol=1e-7)
