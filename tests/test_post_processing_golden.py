"""The CRB-patched post-processing and the strategy <-> caller contract against goldens written by the reference itself
(tests/golden/make_goldens.py:gen_post_processing): Detector3DTemplate.post_processing's 15 record keys per frame
(detector3d_template.py:186-409) and the pickle of Strategy.save_points / save_active_labels (strategy.py:28-38,66-75)."""
import os
import pickle
import types

import numpy as np
import pytest
import torch

from oracle import crb_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
NAMES = ['Car', 'Pedestrian', 'Cyclist']


def _gold():
    return np.load(os.path.join(GOLD, 'ref_post_processing.npz'))


def _ref_pickle():
    with open(os.path.join(GOLD, 'ref_selected_frames_epoch_7_rank_0.pkl'), 'rb') as f:
        return pickle.load(f)


def _stats_rows_from_golden(g):
    """(3, C, 5) rows {num_bbox, n_counted, mean, median, variance} as the device kernel would emit them"""
    rows = np.zeros((3, 3, 5), np.float32)
    for b in range(3):
        rows[b, :, 0] = g['f%d_num_bbox' % b]
        rows[b, :, 1] = g['f%d_mean_points_is_tensor' % b]          # > 0 iff some box of the class owns a point
        rows[b, :, 2] = g['f%d_mean_points' % b]
        rows[b, :, 3] = g['f%d_median_points' % b]
        rows[b, :, 4] = g['f%d_variance_points' % b]
    return rows


def _same_value_kind(got, ref):
    """the reference stores a 0-dim tensor or the python int 0; dtype and value must agree"""
    if torch.is_tensor(ref):
        assert torch.is_tensor(got) and got.dim() == 0 and got.dtype == ref.dtype, (got, ref)
        if ref.dtype.is_floating_point:
            np.testing.assert_allclose(float(got), float(ref), rtol=1e-5)
        else:
            assert int(got) == int(ref)
    else:
        assert not torch.is_tensor(got) and got == ref == 0 and isinstance(got, int), (got, ref)


def _compare_pickles(got, ref):
    assert list(got.keys()) == list(ref.keys())
    assert got['frame_id'] == ref['frame_id']
    for key in ('selected_bbox', 'selected_mean_points', 'selected_median_points', 'selected_variance_points'):
        assert len(got[key]) == len(ref[key])
        for dg, dr in zip(got[key], ref[key]):
            assert list(dg.keys()) == list(dr.keys())
            for name in dr:
                _same_value_kind(dg[name], dr[name])


def _strategy(tmpdir, ids):
    from pcdet.query_strategies.strategy import Strategy
    cfg = types.SimpleNamespace(DATA_CONFIG=types.SimpleNamespace(DATASET='KittiDataset'))
    ds = types.SimpleNamespace(sample_id_list=list(ids), kitti_infos=[{'i': i} for i in ids])
    loader = types.SimpleNamespace(dataset=ds, batch_size=2, num_workers=0)
    model = types.SimpleNamespace(model_cfg=types.SimpleNamespace(
        DENSE_HEAD=types.SimpleNamespace(ANCHOR_GENERATOR_CONFIG=[{'class_name': n} for n in NAMES])))
    return Strategy(model, loader, loader, 0, str(tmpdir), cfg)


def test_strategy_pickle_from_stat_rows_matches_reference_pickle(tmp_path):
    """host half of the contract: record rows -> save_points -> save_active_labels -> the reference's pickle"""
    g = _gold()
    st = _strategy(tmp_path, [str(s) for s in g['frame_ids']])
    st.record_gt_stats(torch.from_numpy(_stats_rows_from_golden(g)))
    assert set(st.bbox_records) == set(str(s) for s in g['frame_ids'])
    st.save_active_labels(selected_frames=['000012', '000010'], cur_epoch=7)
    with open(os.path.join(str(tmp_path), 'selected_frames_epoch_7_rank_0.pkl'), 'rb') as f:
        got = pickle.load(f)
    _compare_pickles(got, _ref_pickle())


def test_record_layout_carries_gt_stats_and_follows_the_model_config():
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    from pcdet.query_strategies import scoring
    L = scoring.RecordLayout.for_model(types.SimpleNamespace(model_cfg=pv_rcnn_cfg().MODEL))
    assert (L.max_box, L.num_class) == (128, 3) and L == scoring.DEFAULT_LAYOUT
    assert L.stride == 2 + 10 * 128 + 15
    S = scoring.RecordLayout.for_model(types.SimpleNamespace(model_cfg=second_cfg().MODEL))
    assert S.max_box == 500                                   # SECOND keeps up to NMS_POST_MAXSIZE boxes, no RoI head
    B, P = 2, 300
    rec = {'entropy': torch.rand(B), 'num': torch.tensor([P, 7]), 'pred_labels': torch.randint(1, 4, (B, P)),
           'density': torch.rand(B, P), 'batch_rcnn_cls': None, 'batch_rcnn_reg': None, 'gt_stats': torch.rand(B, 3, 5)}
    rows = scoring.pack_records(rec, S)
    u = scoring.unpack_records(rows, S)
    assert torch.equal(u['gt_stats'], rec['gt_stats']) and torch.equal(u['labels'][:, :P], rec['pred_labels'])
    with pytest.raises(ValueError):
        scoring.pack_records(rec, L)                          # 300 boxes do not fit the 128-box PV-RCNN layout


def test_oracle_gt_statistics_reproduce_the_reference_golden():
    """oracle/crb_oracle.gt_point_statistics (the checker of the GPU kernel) == what the reference's loop produced"""
    g = _gold()
    pts, gt = g['in_points'], g['in_gt_boxes']
    for b in range(3):
        got = crb_oracle.gt_point_statistics(pts[pts[:, 0] == b][:, 1:4], gt[b], 3)
        np.testing.assert_array_equal([r[0] for r in got], g['f%d_num_bbox' % b])
        np.testing.assert_allclose([r[2] for r in got], g['f%d_mean_points' % b], rtol=1e-6)
        np.testing.assert_allclose([r[3] for r in got], g['f%d_median_points' % b], rtol=0)
        np.testing.assert_allclose([r[4] for r in got], g['f%d_variance_points' % b], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
def _device_batch(g, dev):
    bd = {k[3:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith('in_')}
    off = np.concatenate([[0], np.cumsum(np.bincount(g['in_points'][:, 0].astype(np.int64), minlength=3))])
    bd.update({'batch_size': 3, 'cls_preds_normalized': False, 'has_class_labels': True,
               'point_frame_offsets': torch.from_numpy(off.astype(np.int32)).to(dev),
               'point_frame_counts_host': np.diff(off).tolist()})
    return bd


def _fake_model():
    from pcdet.config import EasyDict
    from golden._constants import POST_CFG
    return types.SimpleNamespace(model_cfg=EasyDict({
        'POST_PROCESSING': POST_CFG, 'DENSE_HEAD': {'ANCHOR_GENERATOR_CONFIG': [{'class_name': n} for n in NAMES]}}))


@pytest.mark.gpu
def test_crb_post_processing_matches_reference_records(dev):
    from pcdet.models.detectors.post_processing import crb_post_processing
    g = _gold()
    pred_dicts, recall = crb_post_processing(_fake_model(), _device_batch(g, dev))
    assert len(pred_dicts) == 3
    ref_keys = {'confidence', 'rpn_preds', 'num_bbox', 'mean_points', 'median_points', 'variance_points',
                'loss_predictions', 'batch_rcnn_cls', 'batch_rcnn_reg', 'embeddings', 'pred_logits', 'pred_boxes',
                'pred_scores', 'pred_labels', 'pred_box_unique_density'}
    for b, d in enumerate(pred_dicts):
        assert ref_keys <= set(d.keys())
        for k, tol in (('pred_boxes', 0), ('pred_labels', 0), ('pred_logits', 0), ('pred_scores', 1e-6),
                       ('pred_box_unique_density', 1e-6), ('batch_rcnn_cls', 1e-6), ('batch_rcnn_reg', 1e-6),
                       ('confidence', 1e-6)):
            ref = g['f%d_%s' % (b, k)]
            got = d[k].cpu().numpy()
            assert got.shape == ref.shape, (b, k, got.shape, ref.shape)
            if tol == 0:
                np.testing.assert_array_equal(got, ref, err_msg='%d %s' % (b, k))
            else:
                np.testing.assert_allclose(got, ref, rtol=tol, atol=tol, err_msg='%d %s' % (b, k))
        for k in ('num_bbox', 'mean_points', 'median_points', 'variance_points'):
            for ci, name in enumerate(NAMES):
                v = d[k][name]
                assert torch.is_tensor(v) == bool(g['f%d_%s_is_tensor' % (b, k)][ci]), (b, k, name, v)
                np.testing.assert_allclose(float(v), g['f%d_%s' % (b, k)][ci], rtol=1e-5)
    assert sorted(recall.keys()) == [str(k) for k in g['recall_keys']]
    assert [recall[str(k)] for k in g['recall_keys']] == g['recall_vals'].tolist()


@pytest.mark.gpu
def test_device_records_to_pickle_matches_reference_pickle(dev, tmp_path):
    """device half + host half: crb_frame_records -> fixed-stride rows -> (all-gather is the identity at world 1) ->
    record_gt_stats -> save_active_labels == the pickle the reference wrote"""
    from pcdet.models.detectors.post_processing import crb_frame_records
    from pcdet.query_strategies import scoring
    g = _gold()
    rows = scoring.pack_records(crb_frame_records(_fake_model(), _device_batch(g, dev)))
    st = _strategy(tmp_path, [str(s) for s in g['frame_ids']])
    st.record_gt_stats(scoring.unpack_records(rows)['gt_stats'])
    st.save_active_labels(selected_frames=['000012', '000010'], cur_epoch=7)
    with open(os.path.join(str(tmp_path), 'selected_frames_epoch_7_rank_0.pkl'), 'rb') as f:
        _compare_pickles(pickle.load(f), _ref_pickle())


@pytest.mark.gpu
@pytest.mark.parametrize('G,n_pts,all_inside', [(12, 4000, False), (300, 6000, False), (5, 400, True), (0, 100, False)])
def test_gt_point_stats_kernel_matches_oracle(dev, G, n_pts, all_inside):
    """crb_gt_point_stats vs the step-by-step restatement of the reference loop, incl. > 256 boxes (LDS chunking), frames
    whose points ALL lie in boxes of a class (the `[1:]` then drops a real box), an empty frame and zero gt boxes"""
    from pcdet.models.detectors.post_processing import gt_point_stats_device
    rng = np.random.default_rng(G + n_pts)
    B = 3
    gt = np.zeros((B, max(G, 1), 8), np.float32)
    pts = []
    for b in range(B):
        n_b = 0 if (b == 1 and not all_inside) else n_pts
        for k in range(G - (b % 2)):                              # ragged: one padded row on odd frames
            gt[b, k] = [rng.uniform(0, 60), rng.uniform(-30, 30), -1, rng.uniform(1, 4), rng.uniform(0.6, 2), 1.6,
                        rng.uniform(-3.1, 3.1), rng.integers(1, 4)]
        if all_inside:
            gt[b, :G, 7] = 1
            k = rng.integers(0, G - (b % 2), n_b)
            loc = rng.uniform(-0.45, 0.45, (n_b, 3)) * gt[b, k, 3:6]
            ca, sa = np.cos(gt[b, k, 6]), np.sin(gt[b, k, 6])
            p = np.stack([loc[:, 0] * ca - loc[:, 1] * sa + gt[b, k, 0], loc[:, 0] * sa + loc[:, 1] * ca + gt[b, k, 1],
                          loc[:, 2] + gt[b, k, 2]], 1)
        else:
            p = np.stack([rng.uniform(0, 60, n_b), rng.uniform(-30, 30, n_b), rng.uniform(-1.7, -0.3, n_b)], 1)
        pts.append(np.concatenate([np.full((n_b, 1), b), p, np.zeros((n_b, 1))], 1).astype(np.float32))
    points = np.concatenate(pts)
    off = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int32)
    bd = {'points': torch.from_numpy(points).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B,
          'point_frame_offsets': torch.from_numpy(off).to(dev)}
    stats, cnt = gt_point_stats_device(bd, 3)
    stats = stats.cpu().numpy()
    for b in range(B):
        ref = crb_oracle.gt_point_statistics(pts[b][:, 1:4], gt[b], 3)
        for c in range(3):
            assert stats[b, c, 0] == ref[c][0] and stats[b, c, 1] == ref[c][1], (b, c, stats[b, c], ref[c])
            np.testing.assert_allclose(stats[b, c, 2:4], ref[c][2:4], rtol=1e-6)
            np.testing.assert_allclose(stats[b, c, 4], ref[c][4], rtol=1e-5)
    if all_inside:
        assert (stats[:, 0, 1] == np.array([min(G, (cnt[b].cpu().numpy() > 0).sum()) - 1 for b in range(B)])).all()
