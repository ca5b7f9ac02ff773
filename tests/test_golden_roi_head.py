"""CPU: second-stage (PV-RCNN RoI head) host logic vs goldens produced by the reference's own modules
(tests/golden/make_goldens.py::gen_roi_head). IoU3D inside the assignment comes from the oracle here (CPU); the GPU
tests run the same checks through the HIP kernels."""
import os

import numpy as np
import pytest
import torch

import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_roi_head.npz')


def _t(a):
    return torch.from_numpy(np.array(a))


@pytest.fixture()
def head(monkeypatch):
    from pcdet.config import EasyDict
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.ops.iou3d_nms import iou3d_nms_utils
    from pcdet.models.roi_heads import PVRCNNHead
    monkeypatch.setattr(iou3d_nms_utils, 'boxes_iou3d_gpu',
                        lambda a, b: torch.from_numpy(oracle.boxes_pairwise(a.numpy(), b.numpy(), 2)))
    cfg = pv_rcnn_cfg().MODEL.ROI_HEAD
    cfg.SHARED_FC, cfg.CLS_FC, cfg.REG_FC = [32, 32], [32, 32], [32, 32]
    cfg.ROI_GRID_POOL = EasyDict({'GRID_SIZE': 2, 'MLPS': [[8, 8], [8, 8]], 'POOL_RADIUS': [0.8, 1.6],
                                  'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool'})
    return PVRCNNHead(input_channels=12, model_cfg=cfg, num_class=1)


def test_same_class_iou_assignment(head):
    g = np.load(G)
    mo, ga = head.proposal_target_layer.max_iou_with_same_class_batched(_t(g['roi_rois']), _t(g['roi_labels']),
                                                                        _t(g['roi_gt']))
    np.testing.assert_array_equal(mo.numpy(), g['roi_max_overlaps'])
    np.testing.assert_array_equal(ga.numpy(), g['roi_gt_assignment'])
    assert (g['roi_max_overlaps'] > 0.55).sum() > 10


def test_targets_losses_grads_decode(head):
    g = np.load(G)
    B, R = g['roi_rois'].shape[:2]
    ptl = head.proposal_target_layer
    ious = _t(g['roi_max_overlaps'])
    gt_of = _t(np.stack([g['roi_gt'][b][g['roi_gt_assignment'][b]] for b in range(B)]))
    ptl.sample_rois_for_rcnn = lambda bd, u=None: (_t(g['roi_rois']), gt_of.clone(), ious, torch.zeros(B, R),
                                                   _t(g['roi_labels']))
    head.train()
    td = head.assign_targets({'batch_size': B})
    np.testing.assert_array_equal(td['reg_valid_mask'].numpy(), g['roi_reg_valid_mask'])
    np.testing.assert_allclose(td['rcnn_cls_labels'].numpy(), g['roi_cls_labels'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(td['gt_of_rois'].numpy(), g['roi_gt_of_rois'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(td['gt_of_rois_src'].numpy(), g['roi_gt_of_rois_src'])
    rcnn_cls = _t(g['roi_rcnn_cls']).requires_grad_(True)
    rcnn_reg = _t(g['roi_rcnn_reg']).requires_grad_(True)
    td['rcnn_cls'], td['rcnn_reg'] = rcnn_cls, rcnn_reg
    head.forward_ret_dict = td
    loss, tb = head.get_loss()
    got = np.array([float(loss.detach()), float(tb['rcnn_loss_cls']), float(tb['rcnn_loss_reg']),
                    float(tb['rcnn_loss_corner'])])
    np.testing.assert_allclose(got, g['roi_loss'], rtol=2e-5)
    loss.backward()
    np.testing.assert_allclose(rcnn_cls.grad.numpy(), g['roi_cls_grad'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(rcnn_reg.grad.numpy(), g['roi_reg_grad'], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(td['rcnn_reg_gt'].numpy(), g['roi_rcnn_reg_gt'], rtol=1e-5, atol=1e-5)
    _, bb = head.generate_predicted_boxes(B, _t(g['roi_rois']), rcnn_cls.detach(), rcnn_reg.detach())
    np.testing.assert_allclose(bb.numpy(), g['roi_decoded'], rtol=1e-5, atol=1e-5)


def test_crb_branch_of_the_losses(head):
    g = np.load(G)
    R = 128
    c1 = _t(g['roi_rcnn_cls'][:R]).requires_grad_(True)
    r1 = _t(g['roi_rcnn_reg'][:R]).requires_grad_(True)
    cls_loss, _ = head.get_box_cls_layer_loss({'rcnn_cls': c1, 'rcnn_cls_labels': _t(g['crb_hyp_cls'])})
    reg_loss = head.get_box_reg_layer_loss({'rcnn_reg': r1, 'reg_sample_targets': _t(g['crb_hyp_reg'])})
    assert reg_loss.shape == (1, R, 7)
    np.testing.assert_allclose([float(cls_loss), float(reg_loss.mean())], g['crb_loss'], rtol=1e-6)
    (cls_loss + reg_loss.mean()).backward()
    np.testing.assert_allclose(c1.grad.numpy(), g['crb_cls_grad'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(r1.grad.numpy(), g['crb_reg_grad'], rtol=1e-5, atol=1e-8)


def test_fc_stack_and_grid_points(head):
    g = np.load(G)
    sd = {k[len('fc_state/'):]: _t(g[k]) for k in g.files if k.startswith('fc_state/')}
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith('roi_grid_pool_layer') for k in missing)
    head.eval()
    pooled = _t(g['fc_pooled'])
    flat = pooled.permute(0, 2, 1).contiguous().view(pooled.shape[0], -1, 1)
    _, cls, reg = head._heads(flat)
    np.testing.assert_allclose(cls.detach().numpy(), g['fc_cls'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(reg.detach().numpy(), g['fc_reg'], rtol=1e-4, atol=1e-5)
    glob, _ = head.get_global_grid_points_of_roi(_t(g['roi_rois']), 2)
    np.testing.assert_allclose(glob.numpy(), g['grid_global'], rtol=1e-5, atol=1e-5)


def test_roi_subsampling_rule():
    """the batched, sync-free sampler follows the reference quotas (proposal_target_layer.py:117-193)"""
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models.roi_heads.target_assigner.proposal_target_layer import ProposalTargetLayer
    ptl = ProposalTargetLayer(pv_rcnn_cfg().MODEL.ROI_HEAD.TARGET_CONFIG)
    g = torch.Generator().manual_seed(0)
    mo = torch.rand((6, 512), generator=g)
    mo[1] = mo[1] * 0.09                      # only easy background
    mo[2] = 0.6 + mo[2] * 0.3                 # only foreground
    mo[3, 5:] = 0.3                           # 5 fg-or-random + hard only
    mo[3, :5] = 0.9
    mo[4] = 0.2 + 0.2 * mo[4]                 # only hard background
    u = (torch.rand((6, 512), generator=g), torch.rand((6, 128), generator=g))
    idx = ptl.subsample_rois_batched(mo, u)
    assert idx.shape == (6, 128)
    s = torch.gather(mo, 1, idx)
    n_fg0 = int((mo[0] >= 0.55).sum())
    assert int((s[0] >= 0.55).sum()) == min(64, n_fg0)
    fg_part = idx[0, :min(64, n_fg0)]
    assert len(set(fg_part.tolist())) == len(fg_part)              # fg without replacement
    n_hard0 = int(((mo[0] < 0.55) & (mo[0] >= 0.1)).sum())
    assert int(((s[0] < 0.55) & (s[0] >= 0.1)).sum()) == min(int(64 * 0.8), n_hard0)
    assert (s[1] < 0.1).all() and (s[2] >= 0.55).all() and (s[4] >= 0.1).all() and (s[4] < 0.55).all()
    assert int((s[3] >= 0.55).sum()) == 5 and int(((s[3] < 0.55) & (s[3] >= 0.1)).sum()) == 123


def test_eval_fast_path_equals_module_path(head):
    """folded Conv1d+BN, the once-only first FC layer on the un-permuted pooled tensor (re-ordered weight) and the MC passes
    batched as rows give the same logits as the plain module path (dropout off)"""
    g = np.load(G)
    sd = {k[len('fc_state/'):]: _t(g[k]) for k in g.files if k.startswith('fc_state/')}
    head.load_state_dict(sd, strict=False)
    head.eval()
    pooled = _t(g['fc_pooled'])
    flat = pooled.permute(0, 2, 1).contiguous().view(pooled.shape[0], -1, 1)
    with torch.no_grad():
        passes = head._heads_eval(pooled, 3)             # takes the pooled (BN, G^3, C) tensor itself
        _, cls, reg = head._heads(flat)
    for _, c, r in passes:
        np.testing.assert_allclose(c.numpy(), cls.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(r.numpy(), reg.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(passes[0][1].numpy(), g['fc_cls'], rtol=1e-4, atol=1e-5)     # and the reference golden
