"""CPU: host-side mirror of the dense half (utils, AnchorHeadSingle + target assigner + losses, BaseBEVBackbone, MeanVFE)
vs golden vectors produced by the reference's own modules (tests/golden/make_goldens.py, committed .npz).
Tolerances: exact for integer labels; 1e-6 abs/rel for fp32 elementwise math; 1e-5 for conv stacks."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    z = np.load(os.path.join(G, name))
    return {k: z[k] for k in z.files}


def _t(a):
    return torch.from_numpy(np.array(a))


def test_utils_match_reference():
    from pcdet.utils import box_coder_utils, box_utils, common_utils, loss_utils
    g = _load('ref_utils.npz')
    coder = box_coder_utils.ResidualCoder()
    enc = coder.encode_torch(_t(g['coder_boxes']), _t(g['coder_anchors']))
    np.testing.assert_allclose(enc.numpy(), g['coder_enc'], rtol=1e-6, atol=1e-6)
    dec = coder.decode_torch(enc, _t(g['coder_anchors']))
    np.testing.assert_allclose(dec.numpy(), g['coder_dec'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(common_utils.limit_period(_t(g['lp_val']), 0.5, np.pi).numpy(), g['lp_a'])
    np.testing.assert_array_equal(common_utils.limit_period(_t(g['lp_val']), 0.0, 2 * np.pi).numpy(), g['lp_b'])
    np.testing.assert_allclose(common_utils.rotate_points_along_z(_t(g['rot_pts']), _t(g['rot_ang'])).numpy(),
                               g['rot_out'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(box_utils.boxes3d_nearest_bev_iou(_t(g['iou_a']), _t(g['iou_b'])).numpy(),
                               g['iou_nearest_bev'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(box_utils.boxes_to_corners_3d(_t(g['iou_a'])).numpy(), g['corners'], rtol=1e-6,
                               atol=1e-6)
    f = loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(_t(g['focal_x']), _t(g['focal_t']),
                                                                         _t(g['focal_w']))
    np.testing.assert_allclose(f.numpy(), g['focal_out'], rtol=1e-6, atol=1e-7)
    s = loss_utils.WeightedSmoothL1Loss(code_weights=[1, 1, 1, 1, 1, 1, 2.0])(_t(g['sl1_p']), _t(g['sl1_q']),
                                                                               _t(g['focal_w']))
    np.testing.assert_allclose(s.numpy(), g['sl1_out'], rtol=1e-6, atol=1e-7)
    c = loss_utils.WeightedCrossEntropyLoss()(_t(g['ce_x']), _t(g['ce_t']), _t(g['focal_w']))
    np.testing.assert_allclose(c.numpy(), g['ce_out'], rtol=1e-6, atol=1e-7)
    cl = loss_utils.get_corner_loss_lidar(_t(g['iou_a'][:30]), _t(g['iou_b']))
    np.testing.assert_allclose(cl.numpy(), g['corner_loss'], rtol=1e-5, atol=1e-6)


def _small_head():
    from pcdet.model_cfgs import second_cfg
    from pcdet.models.dense_heads import AnchorHeadSingle
    cfg = second_cfg().MODEL.DENSE_HEAD
    return AnchorHeadSingle(cfg, input_channels=24, num_class=3, class_names=['Car', 'Pedestrian', 'Cyclist'],
                            grid_size=np.array([176, 160, 40]), point_cloud_range=np.array([0, -8, -3, 17.6, 8, 1],
                                                                                           np.float32),
                            predict_boxes_when_training=True)


def test_anchor_head_targets_losses_grads_match_reference():
    g = _load('ref_anchor_head.npz')
    head = _small_head()
    head.load_state_dict({k[len('head_state/'):]: _t(v) for k, v in g.items() if k.startswith('head_state/')})
    head.train()
    np.testing.assert_array_equal(np.stack([a.numpy() for a in head.anchors]), g['head_anchors'])
    feats = _t(g['head_feats']).requires_grad_(True)
    B = feats.shape[0]
    dd = head({'spatial_features_2d': feats, 'gt_boxes': _t(g['head_gt']), 'batch_size': B})
    fr = head.forward_ret_dict
    np.testing.assert_array_equal(fr['box_cls_labels'].numpy(), g['head_labels'].astype(np.int32))   # bit-exact
    np.testing.assert_allclose(fr['box_reg_targets'].numpy(), g['head_reg_targets'], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(fr['reg_weights'].numpy(), g['head_reg_weights'])
    assert (g['head_labels'] > 0).sum() > 20 and (g['head_labels'] < 0).sum() > 0     # the case is not degenerate
    loss, tb = head.get_loss()
    got = np.array([float(loss), float(tb['rpn_loss_cls']), float(tb['rpn_loss_loc']), float(tb['rpn_loss_dir'])])
    np.testing.assert_allclose(got, g['head_loss'], rtol=2e-6)
    loss.backward()
    np.testing.assert_allclose(feats.grad.numpy(), g['head_feats_grad'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(dd['batch_cls_preds'].detach().numpy(), g['head_batch_cls_preds'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(dd['batch_box_preds'].detach().numpy(), g['head_batch_box_preds'], rtol=1e-5, atol=1e-5)


def test_full_size_anchor_grid_matches_reference():
    from pcdet.model_cfgs import KITTI_ANCHORS
    from pcdet.models.dense_heads.target_assigner.anchor_generator import AnchorGenerator
    g = _load('ref_anchor_head.npz')
    ag = AnchorGenerator(np.array([0, -40, -3, 70.4, 40, 1], np.float32), KITTI_ANCHORS)
    al, per = ag.generate_anchors([np.array([176, 200])] * 3)
    assert list(al[0].shape) == list(g['anchors_full_shape']) and per == [2, 2, 2]
    flat = torch.stack(al).reshape(3, -1, 7)
    np.testing.assert_array_equal(flat[:, ::997].numpy(), g['anchors_full_sample'])
    np.testing.assert_allclose(flat.double().sum(1).numpy(), g['anchors_full_colsum'], rtol=1e-12)


def test_single_frame_entry_point_consistent_with_batched():
    head = _small_head()
    g = _load('ref_anchor_head.npz')
    gt = _t(g['head_gt'])
    full = head.assign_targets(gt)
    ta = head.target_assigner
    anchors = head.anchors[0].reshape(-1, 7)
    m = gt[0, :, 7] == 1
    single = ta.assign_targets_single(anchors, gt[0, m, :7], gt[0, m, 7].int(), 0.6, 0.45)
    lab = full['box_cls_labels'][0].view(1, 20, 22, 3, 2)[..., 0, :].reshape(-1)
    np.testing.assert_array_equal(single['box_cls_labels'].numpy(), lab.numpy())


def test_bev_backbone_and_mean_vfe_match_reference():
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import BaseBEVBackbone
    from pcdet.models.backbones_3d.vfe import MeanVFE
    g = _load('ref_bev_vfe.npz')
    cfg = EasyDict({'LAYER_NUMS': [2, 1], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [8, 16], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [8, 8]})
    m = BaseBEVBackbone(cfg, input_channels=6)
    sd = {k[len('bev_state/'):]: _t(v) for k, v in g.items() if k.startswith('bev_state/')}
    assert set(sd.keys()) == set(m.state_dict().keys())          # same module tree => checkpoints interchange
    m.load_state_dict(sd)
    m.train()
    # golden state was saved AFTER the reference's training-mode forward updated the BN running stats;
    # batch-stat normalisation does not read them, so the outputs are comparable
    y = m({'spatial_features': _t(g['bev_x'])})['spatial_features_2d']
    np.testing.assert_allclose(y.detach().numpy(), g['bev_y'], rtol=1e-4, atol=1e-5)
    vfe = MeanVFE(EasyDict({}), 4)
    out = vfe({'voxels': _t(g['vfe_v']), 'voxel_num_points': _t(g['vfe_n'])})['voxel_features']
    np.testing.assert_allclose(out.numpy(), g['vfe_out'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(out.numpy(), __import__('oracle').mean_vfe(g['vfe_v'], g['vfe_n'].astype(np.int32)),
                               rtol=1e-6, atol=1e-7)


def test_bev_backbone_eval_folding_equals_module_path():
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import BaseBEVBackbone
    cfg = EasyDict({'LAYER_NUMS': [2, 1], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [8, 16], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [8, 8]})
    torch.manual_seed(0)
    m = BaseBEVBackbone(cfg, input_channels=6)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3)
            mod.running_var.uniform_(0.5, 2.0)
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.normal_(0, 0.2)
    m.eval()
    x = torch.randn(2, 6, 20, 24)
    with torch.no_grad():
        a = m({'spatial_features': x})['spatial_features_2d']
    b = m({'spatial_features': x})['spatial_features_2d']          # grad enabled -> module path
    np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-4, atol=1e-5)


def test_fused_head_convs_equal_three_convs():
    """AnchorHeadSingle: one conv over the concatenated cls/box/dir filters == the three module convs, outputs and grads"""
    from pcdet.models.dense_heads import anchor_head_single as A
    from pcdet.models.dense_heads import AnchorHeadSingle
    from pcdet.model_cfgs import second_cfg
    torch.manual_seed(0)
    cfg = second_cfg('kitti').MODEL.DENSE_HEAD
    head = AnchorHeadSingle(cfg, input_channels=32, num_class=3, class_names=['Car', 'Pedestrian', 'Cyclist'],
                            grid_size=np.array([176 * 8, 200 * 8, 40]), point_cloud_range=np.array([0, -40, -3, 70.4, 40, 1.0]),
                            predict_boxes_when_training=False)
    head.eval()
    x1 = torch.randn(2, 32, 200, 176, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    outs = []
    for flag, x in ((True, x1), (False, x2)):
        A.FUSED_HEAD_CONVS = flag
        head.zero_grad()
        d = head({'spatial_features_2d': x, 'batch_size': 2})
        r = head.forward_ret_dict
        loss = r['cls_preds'].square().sum() + r['box_preds'].sum() * 0.5 + r['dir_cls_preds'].abs().sum()
        loss.backward()
        outs.append(([r[k].detach().clone() for k in ('cls_preds', 'box_preds', 'dir_cls_preds')], x.grad.clone(),
                     [p.grad.clone() for p in head.parameters()]))
    A.FUSED_HEAD_CONVS = True
    for a, b in zip(outs[0][0], outs[1][0]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5)
    for a, b in zip(outs[0][2], outs[1][2]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-3)


def test_badge_hypothetical_labels_and_gradient_embedding_match_reference():
    """BadgeSampling's two rules (badge_sampling.py:86-89,146-160) through the product's own helpers against
    ref_badge.npz (reference AnchorHeadSingle, eval pass -> arg-max labels, then per frame get_cls_layer_loss(new_data) ->
    conv_cls.weight.grad): labels exact, embeddings 1e-5 relative to the largest entry of the row."""
    from pcdet.query_strategies.badge_sampling import BadgeSampling
    g = _load('ref_badge.npz')
    head = _small_head()
    head.load_state_dict({k[len('badge_state/'):]: _t(v) for k, v in g.items() if k.startswith('badge_state/')})
    feats = _t(g['badge_feats'])
    B = feats.shape[0]
    head.eval()
    with torch.no_grad():
        dd = head({'spatial_features_2d': feats, 'batch_size': B})
    np.testing.assert_allclose(dd['rpn_preds'].numpy(), g['badge_rpn_preds'], rtol=1e-5, atol=1e-5)
    labels = BadgeSampling.hypothetical_labels(_t(g['badge_rpn_preds']), head.num_class)
    np.testing.assert_array_equal(labels.numpy(), g['badge_labels'].astype(np.int64))
    assert len(np.unique(g['badge_labels'])) == 3
    head.train()
    for b in range(B):
        d1 = head({'spatial_features_2d': feats[b:b + 1], 'gt_boxes': _t(g['badge_gt'][b:b + 1]), 'batch_size': 1})
        emb = BadgeSampling.head_embedding(head, d1['rpn_preds'], labels[b]).numpy()
        ref = g['badge_emb'][b]
        np.testing.assert_allclose(emb, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
        assert np.abs(ref).max() > 1e-3
