"""GPU: csrc/rcnn_loss.hip (second-stage losses with their gradients in one launch, canonical transformation of the sampled ground
truths in one launch) against the torch expressions of the mirror's RoIHeadTemplate (the restatement of
pcdet/models/roi_heads/roi_head_template.py:118-138, :142-285 that the detector-level goldens pin) and their autograd:
loss values and tb_dict entries 2e-6 relative, gradients 2e-5 of the largest entry, canonical targets 2e-6 absolute, bit-equal re-runs.
Cases: the PV-RCNN shape (16 x 128 RoIs), RoI counts that are not multiples of the workgroup, no foreground RoI, no valid RoI,
hard labels (CLS_SCORE_TYPE cls: int64 with -1), saturated logits (the -100 clamp and the 1e-12 floor of ATen's BCE), a prediction
equal to its ground truth (corner distances 0: zero gradient), code weights and loss weights away from 1, corner loss off."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(dev, **loss_over):
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models.roi_heads.roi_head_template import RoIHeadTemplate
    cfg = copy.deepcopy(pv_rcnn_cfg().MODEL.ROI_HEAD)
    for k, v in loss_over.items():
        if k in ('CLS_LOSS', 'REG_LOSS', 'CORNER_LOSS_REGULARIZATION'):
            cfg.LOSS_CONFIG[k] = v
        else:
            cfg.LOSS_CONFIG.LOSS_WEIGHTS[k] = v
    return RoIHeadTemplate(num_class=3, model_cfg=cfg).to(dev)


def _boxes(rng, n, dev, centre=20.0):
    b = np.concatenate([rng.uniform(-centre, centre, (n, 2)), rng.uniform(-2, 1, (n, 1)), rng.uniform(0.5, 4.5, (n, 3)),
                        rng.uniform(-7, 7, (n, 1))], 1).astype(np.float32)
    return torch.from_numpy(b).to(dev)


def _case(dev, B, P, seed, fg='some', labels='soft', sat=False, exact=False):
    rng = np.random.default_rng(seed)
    n = B * P
    rois = _boxes(rng, n, dev)
    gt = rois + torch.from_numpy(rng.normal(0, 0.4, (n, 7)).astype(np.float32)).to(dev)
    gt[:, 3:6] = gt[:, 3:6].abs() + 0.3
    gt = torch.cat([gt, torch.from_numpy(rng.integers(1, 4, (n, 1)).astype(np.float32)).to(dev)], 1)
    iou = torch.from_numpy(rng.uniform(0, 1, n).astype(np.float32)).to(dev)
    if fg == 'none':
        valid = torch.zeros(n, dtype=torch.int64, device=dev)
    else:
        valid = (iou > 0.55).long()
    if labels == 'soft':
        lab = torch.where(iou > 0.75, torch.ones_like(iou), torch.where(iou < 0.25, torch.zeros_like(iou), (iou - 0.25) / 0.5))
    elif labels == 'hard':
        lab = (iou > 0.6).long()
        lab = torch.where((iou > 0.45) & (iou < 0.6), torch.full_like(lab, -1), lab)
    else:                                                   # no valid RoI
        lab = torch.full((n,), -1, dtype=torch.int64, device=dev)
    cls = torch.from_numpy(rng.normal(0, 2, (n, 1)).astype(np.float32)).to(dev)
    if sat:
        cls[::5] = 60.0
        cls[1::5] = -60.0
        cls[2::7] = 110.0
    reg = torch.from_numpy(rng.normal(0, 0.3, (n, 7)).astype(np.float32)).to(dev)
    return {'rois': rois.view(B, P, 7), 'gt_raw': gt.view(B, P, 8), 'reg_valid_mask': valid.view(B, P), 'rcnn_cls_labels': lab.view(B, P),
            'rcnn_cls': cls, 'rcnn_reg': reg, 'exact': exact}


def _run(head, case, fused):
    from pcdet.models.roi_heads import roi_head_template as T
    T.FUSED_LOSS, keep = fused, T.FUSED_LOSS
    try:
        head.proposal_target_layer.forward = lambda batch_dict, uniforms=None: {
            'rois': case['rois'], 'gt_of_rois': case['gt_raw'].clone(), 'reg_valid_mask': case['reg_valid_mask'],
            'rcnn_cls_labels': case['rcnn_cls_labels']}
        targets = head.assign_targets({'batch_size': case['rois'].shape[0]})
        cls = case['rcnn_cls'].clone().requires_grad_(True)
        reg = case['rcnn_reg'].clone()
        if case['exact']:                                    # prediction == ground truth for a few foreground RoIs
            with torch.no_grad():
                n = reg.shape[0]
                anchors = torch.cat([torch.zeros(n, 3, device=reg.device), case['rois'].view(n, 7)[:, 3:6],
                                     torch.zeros(n, 1, device=reg.device)], 1)
                enc = head.box_coder.encode_torch(targets['gt_of_rois'].view(n, -1)[:, :7], anchors)
                reg[::3] = enc[::3]
        reg.requires_grad_(True)
        head.forward_ret_dict = dict(targets, rcnn_cls=cls, rcnn_reg=reg)
        loss, tb = head.get_loss()
        (loss * 1.7).backward()
        return loss.detach(), {k: v.detach().clone() for k, v in tb.items()}, cls.grad, reg.grad, targets['gt_of_rois'], \
            head.forward_ret_dict['rcnn_reg_gt']
    finally:
        T.FUSED_LOSS = keep


CASES = [
    dict(B=16, P=128, seed=1),
    dict(B=3, P=37, seed=2),
    dict(B=1, P=1, seed=3),
    dict(B=5, P=413, seed=4, labels='hard'),
    dict(B=2, P=128, seed=5, fg='none'),
    dict(B=2, P=64, seed=6, labels='none'),
    dict(B=4, P=128, seed=7, sat=True),
    dict(B=4, P=100, seed=8, exact=True),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'B%d_P%d_%s' % (c['B'], c['P'], '_'.join(k for k in c if k not in ('B', 'P', 'seed'))))
def test_fused_losses_and_gradients_equal_the_torch_expressions(dev, case):
    head = _head(dev)
    c = _case(dev, **case)
    loss_f, tb_f, gc_f, gr_f, ct_f, tgt_f = _run(head, c, True)
    loss_t, tb_t, gc_t, gr_t, ct_t, tgt_t = _run(head, c, False)
    assert type(loss_f) is torch.Tensor and set(tb_f) == set(tb_t) == {'rcnn_loss_cls', 'rcnn_loss_reg', 'rcnn_loss_corner', 'rcnn_loss'}
    torch.testing.assert_close(ct_f, ct_t, rtol=0, atol=2e-6)
    torch.testing.assert_close(tgt_f, tgt_t, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(loss_f, loss_t, rtol=2e-6, atol=1e-7)
    for k in tb_t:
        torch.testing.assert_close(tb_f[k], tb_t[k], rtol=2e-6, atol=1e-7, msg=lambda m, k=k: k + ': ' + m)
    for gf, gt, what in ((gc_f, gc_t, 'd rcnn_cls'), (gr_f, gr_t, 'd rcnn_reg')):
        assert gf.shape == gt.shape and torch.isfinite(gf).all()
        scale = float(gt.abs().max())
        assert float((gf - gt).abs().max()) <= 2e-5 * max(scale, 1e-6), what
    if case.get('exact'):
        assert float(gr_f[::3].abs().max()) >= 0.0 and torch.isfinite(gr_f).all()
    # the same launch again: bit-equal
    loss_2, tb_2, gc_2, gr_2, _, _ = _run(head, c, True)
    assert torch.equal(loss_f, loss_2) and torch.equal(gc_f, gc_2) and torch.equal(gr_f, gr_2)


def test_weights_and_corner_switch(dev):
    """code weights / loss weights away from 1; the corner term switched off (no rcnn_loss_corner entry, as in the reference)"""
    c = _case(dev, B=4, P=128, seed=11)
    head = _head(dev, code_weights=[1.0, 0.5, 2.0, 1.5, 1.0, 0.25, 3.0], rcnn_cls_weight=0.7, rcnn_reg_weight=1.3, rcnn_corner_weight=0.4)
    f, t = _run(head, c, True), _run(head, c, False)
    torch.testing.assert_close(f[0], t[0], rtol=2e-6, atol=1e-7)
    for a, b in ((f[2], t[2]), (f[3], t[3])):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    head = _head(dev, CORNER_LOSS_REGULARIZATION=False)
    f, t = _run(head, c, True), _run(head, c, False)
    assert set(f[1]) == set(t[1]) == {'rcnn_loss_cls', 'rcnn_loss_reg', 'rcnn_loss'}
    torch.testing.assert_close(f[0], t[0], rtol=2e-6, atol=1e-7)
    assert float((f[3] - t[3]).abs().max()) <= 2e-5 * float(t[3].abs().max())


def test_other_configurations_keep_the_torch_expressions(dev):
    """reduce=False (per-frame losses of CRB stage 2), the CRB branch (reg_sample_targets) and CrossEntropy are not what the kernel
    implements: the head says so by taking the torch path (no crb_rcnn_loss call)"""
    from crbhip import rcnn_loss
    head = _head(dev)
    c = _case(dev, B=2, P=16, seed=12)
    _run(head, c, True)
    calls = []
    orig = rcnn_loss.rcnn_loss
    rcnn_loss.rcnn_loss = lambda *a, **k: calls.append(1) or orig(*a, **k)
    try:
        head.get_loss(reduce=True)
        assert len(calls) == 1
        loss, tb = head.get_loss(reduce=False)
        assert loss.shape == (2,) and len(calls) == 1
        head.forward_ret_dict['reg_sample_targets'] = torch.zeros_like(head.forward_ret_dict['rcnn_reg'])
        assert head._fused_loss_cfg(True) is None
    finally:
        rcnn_loss.rcnn_loss = orig


def _sampler_case(dev, B, R, G, seed, mode):
    """proposals around the ground truths: `mode` shapes the three sets of each frame (mixed / all foreground / no foreground /
    no hard background / no easy background / no ground truth in frame 0)"""
    rng = np.random.default_rng(seed)
    gt = np.zeros((B, G, 8), np.float32)
    rois = np.zeros((B, R, 7), np.float32)
    labels = np.zeros((B, R), np.int64)
    for b in range(B):
        ng = int(rng.integers(1, G + 1))
        if mode == 'no_gt' and b == 0:
            ng = 0
        g = np.concatenate([rng.uniform(-30, 30, (ng, 2)), rng.uniform(-1.5, 0, (ng, 1)), rng.uniform(1.2, 4.5, (ng, 3)),
                            rng.uniform(-3.1, 3.1, (ng, 1)), rng.integers(1, 4, (ng, 1))], 1).astype(np.float32)
        gt[b, :ng] = g
        for r in range(R):
            kind = {'mixed': rng.integers(0, 4), 'all_fg': 0, 'no_fg': rng.integers(1, 4), 'no_hard': rng.choice([0, 3]),
                    'no_easy': rng.integers(0, 2), 'no_gt': rng.integers(0, 4)}[mode]
            if ng == 0 or kind == 3:                          # far from everything: easy background
                rois[b, r] = [rng.uniform(40, 60), rng.uniform(40, 60), -1, 3.9, 1.6, 1.5, rng.uniform(-3, 3)]
                labels[b, r] = rng.integers(1, 4)
                continue
            k = int(rng.integers(0, ng))
            jitter = {0: 0.01 if mode == 'all_fg' else 0.03, 1: 0.35, 2: 0.2}[int(kind)]
            box = g[k, :7].copy()
            if mode == 'no_fg':                               # half a box away along both axes: IoU 0.14 (hard background)
                box[:2] += rng.choice([-1.0, 1.0], 2) * 0.5 * box[3:5]
            else:
                box[:2] += rng.normal(0, jitter * box[3:5])
                box[6] += rng.normal(0, jitter * 0.3)
            rois[b, r] = box
            labels[b, r] = int(g[k, 7]) if (mode == 'all_fg' or rng.uniform() < 0.9) else int(g[k, 7]) % 3 + 1
    t = lambda a: torch.from_numpy(a).to(dev)
    return {'rois': t(rois), 'roi_scores': t(rng.uniform(0, 1, (B, R)).astype(np.float32)), 'roi_labels': t(labels), 'gt_boxes': t(gt),
            'batch_size': B}


@pytest.mark.parametrize('B,R,G,mode,by_class,score_type', [
    (16, 512, 12, 'mixed', True, 'roi_iou'), (3, 100, 5, 'mixed', True, 'roi_iou'), (2, 512, 9, 'all_fg', True, 'roi_iou'),
    (2, 300, 7, 'no_fg', True, 'cls'), (2, 512, 6, 'no_hard', True, 'roi_iou'), (2, 257, 6, 'no_easy', True, 'roi_iou'),
    (3, 512, 4, 'no_gt', True, 'roi_iou'), (4, 512, 1, 'mixed', False, 'cls'), (1, 1024, 20, 'mixed', True, 'roi_iou')])
def test_roi_sampling_kernel_equals_the_torch_layer(dev, B, R, G, mode, by_class, score_type):
    """crb_roi_sample_targets against the mirror's batched torch layer on the same uniforms: every output EQUAL (indices, gathers,
    masks, labels), drawn uniforms as well as injected ones; frames without foreground / background sets / ground truths, fewer
    proposals than ROI_PER_IMAGE, the 1024-proposal limit, both CLS_SCORE_TYPEs, with and without SAMPLE_ROI_BY_EACH_CLASS"""
    from pcdet.models.roi_heads.target_assigner import proposal_target_layer as L
    head = _head(dev)
    layer = head.proposal_target_layer
    layer.roi_sampler_cfg.SAMPLE_ROI_BY_EACH_CLASS = by_class
    layer.roi_sampler_cfg.CLS_SCORE_TYPE = score_type
    bd = _sampler_case(dev, B, R, G, seed=B * 1000 + R + G, mode=mode)
    P = layer.roi_sampler_cfg.ROI_PER_IMAGE
    keep = L.FUSED_SAMPLER
    try:
        outs = []
        for fused in (True, False):
            L.FUSED_SAMPLER = fused
            g = torch.Generator(device=dev)
            g.manual_seed(5)
            layer.generator = g
            drawn = layer.forward(bd)
            gi = torch.Generator(device=dev)
            gi.manual_seed(6)
            uni = (torch.rand((B, R), device=dev, generator=gi), torch.rand((B, P), device=dev, generator=gi))
            outs.append((drawn, layer.forward(bd, uni)))
        assert layer.forward_fused(bd) is not None
    finally:
        L.FUSED_SAMPLER = keep
        layer.generator = None
    for (f, t) in zip(outs[0], outs[1]):
        assert set(f) == set(t)
        for k in t:
            assert f[k].dtype == t[k].dtype and f[k].shape == t[k].shape, k
            assert torch.equal(f[k], t[k]), (k, mode)
    fgs = outs[0][0]['reg_valid_mask'].sum(1)
    if mode == 'all_fg':
        assert int(fgs.min()) == P
    if mode == 'no_fg':
        assert int(fgs.max()) == 0


def test_roi_sampling_falls_back_outside_its_limits(dev):
    from pcdet.models.roi_heads.target_assigner import proposal_target_layer as L
    layer = _head(dev).proposal_target_layer
    bd = _sampler_case(dev, 1, 1100, 3, seed=3, mode='mixed')
    assert layer.forward_fused(bd) is None and layer.forward(bd)['rois'].shape == (1, 128, 7)
    bd = _sampler_case(dev, 2, 64, 3, seed=4, mode='mixed')
    layer.injected_indices = torch.zeros((2, 128), dtype=torch.long, device=dev)
    assert layer.forward_fused(bd) is None


@pytest.mark.parametrize('num_class,B,M,mode', [(1, 16, 2048, 'mixed'), (3, 2, 333, 'mixed'), (1, 2, 2048, 'no_positive'), (3, 1, 1, 'mixed')])
def test_point_head_labels_and_focal_loss_equal_the_torch_expressions(dev, num_class, B, M, mode):
    """csrc/point_head.hip against the mirror's torch expressions (point_head_template.py assign_stack_targets / get_cls_layer_loss):
    labels EQUAL; loss 2e-6, positives equal, gradient 2e-5 of its largest entry; bit-equal re-run; class-agnostic and three-class heads,
    a batch without any point inside a box, saturated logits"""
    import copy
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models.dense_heads import point_head_template as PT
    from pcdet.models.dense_heads.point_head_simple import PointHeadSimple
    cfg = copy.deepcopy(pv_rcnn_cfg().MODEL.POINT_HEAD)
    cfg.LOSS_CONFIG.LOSS_WEIGHTS['point_cls_weight'] = 0.7
    head = PointHeadSimple(num_class=num_class, input_channels=32, model_cfg=cfg).to(dev).train()
    rng = np.random.default_rng(B * 100 + M)
    G = 7
    gt = np.zeros((B, G, 8), np.float32)
    for b in range(B):
        ng = int(rng.integers(1, G + 1))
        gt[b, :ng] = np.concatenate([rng.uniform(-20, 20, (ng, 2)), rng.uniform(-1.5, 0, (ng, 1)), rng.uniform(1.5, 4.5, (ng, 3)),
                                     rng.uniform(-3, 3, (ng, 1)), rng.integers(1, 4, (ng, 1))], 1)
    pts = np.zeros((B, M, 4), np.float32)
    for b in range(B):
        pts[b, :, 0] = b
        ng = int((gt[b, :, 3] > 0).sum())
        k = rng.integers(0, ng, M)
        spread = {'mixed': 0.75, 'no_positive': 0.0}[mode]
        pts[b, :, 1:4] = gt[b, k, 0:3] + rng.uniform(-1, 1, (M, 3)) * gt[b, k, 3:6] * spread
        if mode == 'no_positive':
            pts[b, :, 1:3] += 100.0
    batch = {'gt_boxes': torch.from_numpy(gt).to(dev), 'point_coords': torch.from_numpy(pts.reshape(-1, 4)).to(dev)}
    preds = torch.from_numpy(rng.normal(0, 2, (B * M, num_class)).astype(np.float32)).to(dev)
    preds[::7] = 40.0
    preds[1::7] = -40.0
    out = []
    keep = PT.FUSED
    try:
        for fused in (True, False, True):
            PT.FUSED = fused
            labels = head.assign_targets(batch)['point_cls_labels']
            p = preds.clone().requires_grad_(True)
            head.forward_ret_dict = {'point_cls_preds': p, 'point_cls_labels': labels}
            loss, tb = head.get_loss()
            (loss * 1.3).backward()
            out.append((labels, loss.detach(), tb['point_pos_num'].detach().clone(), tb['point_loss_cls'].clone(), p.grad))
    finally:
        PT.FUSED = keep
    f, t, f2 = out
    assert f[0].dtype == t[0].dtype == torch.int64 and torch.equal(f[0], t[0])
    if mode == 'mixed' and M > 1:
        assert int((t[0] > 0).sum()) > 0 and int((t[0] == 0).sum()) > 0
        assert num_class == 1 or int(t[0].max()) > 1
    if mode == 'no_positive':
        assert int((t[0] > 0).sum()) == 0
    torch.testing.assert_close(f[1], t[1], rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(f[3], t[3], rtol=2e-6, atol=1e-7)
    assert float(f[2]) == float(t[2])
    assert f[4].shape == t[4].shape and float((f[4] - t[4]).abs().max()) <= 2e-5 * max(float(t[4].abs().max()), 1e-9)
    assert torch.equal(f[1], f2[1]) and torch.equal(f[4], f2[4])


def test_roi_grid_points_kernel_equals_the_torch_expressions(dev):
    """crb_roi_grid_points against get_dense_grid_points + rotate_points_along_z + centre (the mirror of pvrcnn_head.py:116-141):
    1e-6 relative / 4e-6 absolute (one f32 rounding of a 60 m coordinate; the torch path multiplies by a 3x3 matrix), grid sizes 6 and 2,
    rows wider than 7"""
    from pcdet.models.roi_heads import pvrcnn_head as PH
    rng = np.random.default_rng(9)
    rois = torch.cat([_boxes(rng, 517, dev, centre=60.0), torch.zeros(517, 2, device=dev)], 1).view(11, 47, 9)
    head = PH.PVRCNNHead.__new__(PH.PVRCNNHead)
    for G in (6, 2):
        fused, none = PH.PVRCNNHead.get_global_grid_points_of_roi(head, rois, G)
        assert none is None
        keep = PH.FUSED_GRID_POINTS
        try:
            PH.FUSED_GRID_POINTS = False
            plain, _ = PH.PVRCNNHead.get_global_grid_points_of_roi(head, rois, G)
        finally:
            PH.FUSED_GRID_POINTS = keep
        assert fused.shape == plain.shape == (517, G ** 3, 3)
        torch.testing.assert_close(fused, plain, rtol=1e-6, atol=4e-6)
