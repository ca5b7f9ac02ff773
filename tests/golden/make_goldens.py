"""Generate golden vectors by importing the importable half of the reference (SURVEY §8c stub recipe).

Runs ONLY in the authoring container (needs /root/reference); the .npz files it writes are committed and are the only
thing that travels.  Usage:  python tests/golden/make_goldens.py
Nothing from the reference is copied: the script imports its modules, feeds seeded inputs, stores inputs + outputs."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _constants import (EasyDict, small_head_cfg, seeded_state, parta2_cfg, parta2_inputs, PP_PCR, PP_VOXEL, PP_KEYPOINTS,  # noqa: E402,F401
                        PP_LEVELS, point_path_inputs, POST_CFG, PV_KEYPOINTS, PV_POINTS, PV_FIRST_FRAME, PV_KINDS, PV_GRADS, pv_grads,
                        PV_SMALL, pv_seeded_state)

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    assert os.path.isdir(REF), 'reference tree not mounted'
    sys.path.insert(0, REF)
    _stub('pcdet.version', __version__='0.5.2+ref')
    _stub('easydict', EasyDict=EasyDict)
    for n in ('SharedArray', 'wandb', 'tensorboardX', 'kornia'):
        _stub(n)
    _passthrough = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f))
    nb = _stub('numba', jit=_passthrough, njit=_passthrough)
    nb.cuda = _stub('numba.cuda', jit=_passthrough)
    _stub('tensorboardX', SummaryWriter=object)
    sk = _stub('skimage')
    sk.transform = _stub('skimage.transform')
    _stub('skimage.io')
    sp = _stub('spconv')
    spp = _stub('spconv.pytorch', SparseModule=torch.nn.Module, SparseSequential=torch.nn.Sequential)
    spp.conv = _stub('spconv.pytorch.conv', SparseConvolution=torch.nn.Module)
    sp.pytorch = spp
    for ext in ('pcdet.ops.iou3d_nms.iou3d_nms_cuda', 'pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda',
                'pcdet.ops.roipoint_pool3d.roipoint_pool3d_cuda',
                'pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda',
                'pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda'):
        _stub(ext)
    # the reference calls .cuda() in constructors; run them on the CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def _np(t):
    return t.detach().cpu().numpy()


def rand_gt(rng, B, G, rng_xy):
    sizes = {1: (3.9, 1.6, 1.56), 2: (0.8, 0.6, 1.73), 3: (1.76, 0.6, 1.73)}
    gt = np.zeros((B, G, 8), np.float32)
    for b in range(B):
        n = G if b % 2 == 0 else G - 3          # ragged: trailing zero rows
        for g in range(n):
            c = int(rng.integers(1, 4))
            s = np.array(sizes[c]) * rng.uniform(0.85, 1.15, 3)
            gt[b, g] = [rng.uniform(*rng_xy[0]), rng.uniform(*rng_xy[1]), -1.0 + rng.uniform(-0.3, 0.3), *s,
                        rng.uniform(-np.pi, np.pi), c]
    return gt


def gen_utils(out):
    from pcdet.utils import box_coder_utils, box_utils, common_utils, loss_utils
    rng = np.random.default_rng(1)
    boxes = rand_gt(rng, 1, 40, ((0, 70), (-40, 40)))[0, :, :7]
    anchors = boxes + rng.normal(0, 0.3, boxes.shape).astype(np.float32)
    anchors[:, 3:6] = np.abs(anchors[:, 3:6]) + 0.1
    coder = box_coder_utils.ResidualCoder()
    enc = coder.encode_torch(torch.from_numpy(boxes.copy()), torch.from_numpy(anchors.copy()))
    dec = coder.decode_torch(enc, torch.from_numpy(anchors.copy()))
    out['coder_boxes'], out['coder_anchors'], out['coder_enc'], out['coder_dec'] = boxes, anchors, _np(enc), _np(dec)
    val = rng.uniform(-20, 20, 200).astype(np.float32)
    out['lp_val'] = val
    out['lp_a'] = _np(common_utils.limit_period(torch.from_numpy(val), offset=0.5, period=np.pi))
    out['lp_b'] = _np(common_utils.limit_period(torch.from_numpy(val), offset=0.0, period=2 * np.pi))
    pts = rng.normal(size=(3, 17, 5)).astype(np.float32)
    ang = rng.uniform(-4, 4, 3).astype(np.float32)
    out['rot_pts'], out['rot_ang'] = pts, ang
    out['rot_out'] = _np(common_utils.rotate_points_along_z(torch.from_numpy(pts), torch.from_numpy(ang)))
    b2 = rand_gt(rng, 1, 30, ((0, 30), (-15, 15)))[0, :, :7]
    out['iou_a'], out['iou_b'] = boxes, b2
    out['iou_nearest_bev'] = _np(box_utils.boxes3d_nearest_bev_iou(torch.from_numpy(boxes), torch.from_numpy(b2)))
    out['corners'] = _np(box_utils.boxes_to_corners_3d(torch.from_numpy(boxes)))
    # losses
    x = rng.normal(0, 2, (2, 50, 3)).astype(np.float32)
    t = (rng.uniform(size=(2, 50, 3)) < 0.2).astype(np.float32)
    w = rng.uniform(0, 1, (2, 50)).astype(np.float32)
    out['focal_x'], out['focal_t'], out['focal_w'] = x, t, w
    out['focal_out'] = _np(loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0)(
        torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(w)))
    p = rng.normal(0, 1, (2, 50, 7)).astype(np.float32)
    q = rng.normal(0, 1, (2, 50, 7)).astype(np.float32)
    q[0, 3, 2] = np.nan
    out['sl1_p'], out['sl1_q'] = p, q
    out['sl1_out'] = _np(loss_utils.WeightedSmoothL1Loss(code_weights=[1, 1, 1, 1, 1, 1, 2.0])(
        torch.from_numpy(p), torch.from_numpy(q), torch.from_numpy(w)))
    logits = rng.normal(0, 1, (2, 50, 2)).astype(np.float32)
    oh = np.eye(2, dtype=np.float32)[rng.integers(0, 2, (2, 50))]
    out['ce_x'], out['ce_t'] = logits, oh
    out['ce_out'] = _np(loss_utils.WeightedCrossEntropyLoss()(torch.from_numpy(logits), torch.from_numpy(oh),
                                                               torch.from_numpy(w)))
    out['corner_loss'] = _np(loss_utils.get_corner_loss_lidar(torch.from_numpy(boxes[:30]), torch.from_numpy(b2)))


def gen_head(out):
    """AnchorHeadSingle on a reduced grid (feature map 22x20): anchors, targets, losses + grads, decoded boxes"""
    from pcdet.models.dense_heads.anchor_head_single import AnchorHeadSingle
    rng = np.random.default_rng(2)
    pc_range = np.array([0, -8, -3, 17.6, 8, 1], np.float32)
    grid = np.array([176, 160, 40], np.int64)
    torch.manual_seed(3)
    head = AnchorHeadSingle(small_head_cfg(), input_channels=24, num_class=3,
                            class_names=['Car', 'Pedestrian', 'Cyclist'], grid_size=grid, point_cloud_range=pc_range,
                            predict_boxes_when_training=True)
    head.train()
    B = 3
    gt = rand_gt(rng, B, 9, ((0.5, 17), (-7.5, 7.5)))
    gt[2] = 0                                   # a frame without any box
    feats = torch.from_numpy(rng.normal(0, 1, (B, 24, 20, 22)).astype(np.float32)).requires_grad_(True)
    dd = head({'spatial_features_2d': feats, 'gt_boxes': torch.from_numpy(gt.copy()), 'batch_size': B})
    loss, tb = head.get_loss()
    loss.backward()
    out['head_state'] = {k: _np(v) for k, v in head.state_dict().items()}
    out['head_feats'], out['head_gt'] = _np(feats), gt
    out['head_anchors'] = np.stack([_np(a) for a in head.anchors])
    fr = head.forward_ret_dict
    out['head_labels'] = _np(fr['box_cls_labels']).astype(np.int8)
    out['head_reg_targets'] = _np(fr['box_reg_targets'])
    out['head_reg_weights'] = _np(fr['reg_weights'])
    out['head_loss'] = np.array([float(loss), tb['rpn_loss_cls'], tb['rpn_loss_loc'], tb['rpn_loss_dir']], np.float64)
    out['head_feats_grad'] = _np(feats.grad)
    out['head_batch_cls_preds'] = _np(dd['batch_cls_preds'])
    out['head_batch_box_preds'] = _np(dd['batch_box_preds'])
    # full-size anchors: a strided sample + column sums
    from pcdet.models.dense_heads.target_assigner.anchor_generator import AnchorGenerator
    cfgs = small_head_cfg().ANCHOR_GENERATOR_CONFIG
    ag = AnchorGenerator(anchor_range=np.array([0, -40, -3, 70.4, 40, 1], np.float32), anchor_generator_config=cfgs)
    al, per = ag.generate_anchors([np.array([176, 200])] * 3)
    out['anchors_full_shape'] = np.array(al[0].shape)
    flat = torch.stack(al).reshape(3, -1, 7)
    out['anchors_full_sample'] = _np(flat[:, ::997])
    out['anchors_full_colsum'] = _np(flat.double().sum(1))


def gen_bev(out):
    from pcdet.models.backbones_2d.base_bev_backbone import BaseBEVBackbone
    from pcdet.models.backbones_3d.vfe.mean_vfe import MeanVFE
    cfg = EasyDict({'LAYER_NUMS': [2, 1], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [8, 16], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [8, 8]})
    torch.manual_seed(4)
    m = BaseBEVBackbone(cfg, input_channels=6)
    m.train()
    rng = np.random.default_rng(5)
    x = rng.normal(0, 1, (2, 6, 20, 24)).astype(np.float32)
    y = m({'spatial_features': torch.from_numpy(x)})['spatial_features_2d']
    out['bev_state'] = {k: _np(v) for k, v in m.state_dict().items()}
    out['bev_x'], out['bev_y'] = x, _np(y)
    v = rng.normal(0, 1, (50, 5, 4)).astype(np.float32)
    n = rng.integers(0, 6, 50).astype(np.float32)
    for i in range(50):
        v[i, int(n[i]):] = 0
    vfe = MeanVFE(EasyDict({}), 4)
    out['vfe_v'], out['vfe_n'] = v, n
    out['vfe_out'] = _np(vfe({'voxels': torch.from_numpy(v), 'voxel_num_points': torch.from_numpy(n)})['voxel_features'])


def gen_roi_head(out):
    """second-stage pieces of PV-RCNN from the reference (CPU): canonical target transform, soft cls labels, losses +
    grads, box decoding, FC stack in eval mode, same-class IoU assignment (IoU3D supplied by oracle/_ref-pinned C code)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))          # repo root for `oracle`
    sys.path.insert(0, os.path.dirname(OUT))
    import oracle
    from boxes_synth import detection_boxes
    from pcdet.config import cfg as ref_cfg
    ref_cfg.CLASS_NAMES = ['Car', 'Pedestrian', 'Cyclist']
    from pcdet.ops.iou3d_nms import iou3d_nms_utils
    iou3d_nms_utils.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(oracle.boxes_pairwise(a.numpy(), b.numpy(), 2))
    from pcdet.models.roi_heads.pvrcnn_head import PVRCNNHead
    head_cfg = EasyDict({
        'NAME': 'PVRCNNHead', 'CLASS_AGNOSTIC': True, 'SAMPLING_ROUND': 5, 'SHARED_FC': [32, 32], 'CLS_FC': [32, 32],
        'REG_FC': [32, 32], 'DP_RATIO': 0.3,
        'NMS_CONFIG': {'TRAIN': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 9000,
                                 'NMS_POST_MAXSIZE': 512, 'NMS_THRESH': 0.8},
                       'TEST': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 1024,
                                'NMS_POST_MAXSIZE': 128, 'NMS_THRESH': 0.7}},
        'ROI_GRID_POOL': {'GRID_SIZE': 2, 'MLPS': [[8, 8], [8, 8]], 'POOL_RADIUS': [0.8, 1.6], 'NSAMPLE': [16, 16],
                          'POOL_METHOD': 'max_pool'},
        'TARGET_CONFIG': {'BOX_CODER': 'ResidualCoder', 'ROI_PER_IMAGE': 128, 'FG_RATIO': 0.5,
                          'SAMPLE_ROI_BY_EACH_CLASS': True, 'CLS_SCORE_TYPE': 'roi_iou', 'CLS_FG_THRESH': 0.75,
                          'CLS_BG_THRESH': 0.25, 'CLS_BG_THRESH_LO': 0.1, 'HARD_BG_RATIO': 0.8, 'REG_FG_THRESH': 0.55},
        'LOSS_CONFIG': {'CLS_LOSS': 'BinaryCrossEntropy', 'REG_LOSS': 'smooth-l1', 'CORNER_LOSS_REGULARIZATION': True,
                        'LOSS_WEIGHTS': {'rcnn_cls_weight': 1.0, 'rcnn_reg_weight': 1.0, 'rcnn_corner_weight': 1.0,
                                         'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}})
    torch.manual_seed(11)
    head = PVRCNNHead(input_channels=12, model_cfg=head_cfg, num_class=1)
    rng = np.random.default_rng(12)
    B, R = 2, 128
    gtb = np.zeros((B, 9, 8), np.float32)
    rois = np.zeros((B, R, 7), np.float32)
    for b in range(B):
        g, _ = detection_boxes(rng, 9 if b == 0 else 6, n_obj=9, spread=0.0)
        gtb[b, :len(g), :7] = g
        gtb[b, :len(g), 7] = rng.integers(1, 4, len(g))
        k = rng.integers(0, len(g), R)
        rois[b] = g[k] + rng.normal(0, 0.25, (R, 7)).astype(np.float32) * np.array([1, 1, .3, .3, .3, .3, .5], np.float32)
    roi_labels = rng.integers(1, 4, (B, R)).astype(np.int64)
    # (e) same-class max IoU
    mo, ga = [], []
    for b in range(B):
        n_gt = 9 if b == 0 else 6
        m, a = head.proposal_target_layer.get_max_iou_with_same_class(
            rois=torch.from_numpy(rois[b]), roi_labels=torch.from_numpy(roi_labels[b]),
            gt_boxes=torch.from_numpy(gtb[b, :n_gt, :7]), gt_labels=torch.from_numpy(gtb[b, :n_gt, 7]).long())
        mo.append(_np(m)); ga.append(_np(a))
    out['roi_rois'], out['roi_gt'], out['roi_labels'] = rois, gtb, roi_labels
    out['roi_max_overlaps'], out['roi_gt_assignment'] = np.stack(mo), np.stack(ga)
    # (d) labels from IoUs with the sampler bypassed
    ious = torch.from_numpy(np.stack(mo))
    gt_of = torch.from_numpy(np.stack([gtb[b][ga[b]] for b in range(B)]))
    ptl = head.proposal_target_layer
    ptl.sample_rois_for_rcnn = lambda batch_dict: (torch.from_numpy(rois), gt_of.clone(), ious,
                                                   torch.zeros(B, R), torch.from_numpy(roi_labels))
    head.train()
    td = head.assign_targets({'batch_size': B})
    out['roi_reg_valid_mask'] = _np(td['reg_valid_mask'])
    out['roi_cls_labels'] = _np(td['rcnn_cls_labels'])
    out['roi_gt_of_rois'] = _np(td['gt_of_rois'])
    out['roi_gt_of_rois_src'] = _np(td['gt_of_rois_src'])
    # (b) losses + grads
    rcnn_cls = torch.from_numpy(rng.normal(0, 1, (B * R, 1)).astype(np.float32)).requires_grad_(True)
    rcnn_reg = torch.from_numpy(rng.normal(0, 0.3, (B * R, 7)).astype(np.float32)).requires_grad_(True)
    td['rcnn_cls'], td['rcnn_reg'] = rcnn_cls, rcnn_reg
    head.forward_ret_dict = td
    loss, tb = head.get_loss()
    loss.backward()
    out['roi_rcnn_cls'], out['roi_rcnn_reg'] = _np(rcnn_cls), _np(rcnn_reg)
    out['roi_loss'] = np.array([float(loss), tb['rcnn_loss_cls'], tb['rcnn_loss_reg'], tb['rcnn_loss_corner']])
    out['roi_cls_grad'], out['roi_reg_grad'] = _np(rcnn_cls.grad), _np(rcnn_reg.grad)
    out['roi_rcnn_reg_gt'] = _np(td['rcnn_reg_gt'])
    # CRB branch of the two losses (crb_sampling.py:194-196)
    hyp_cls = torch.from_numpy(rng.uniform(0, 1, (R, 1)).astype(np.float32))
    hyp_reg = torch.from_numpy(rng.normal(0, 0.3, (R, 7)).astype(np.float32))
    c1 = rcnn_cls[:R].detach().clone().requires_grad_(True)
    r1 = rcnn_reg[:R].detach().clone().requires_grad_(True)
    cls_loss, _ = head.get_box_cls_layer_loss({'rcnn_cls': c1, 'rcnn_cls_labels': hyp_cls})
    reg_loss = head.get_box_reg_layer_loss({'rcnn_reg': r1, 'reg_sample_targets': hyp_reg})
    (cls_loss + reg_loss.mean()).backward()
    out['crb_hyp_cls'], out['crb_hyp_reg'] = _np(hyp_cls), _np(hyp_reg)
    out['crb_loss'] = np.array([float(cls_loss), float(reg_loss.mean())])
    out['crb_cls_grad'], out['crb_reg_grad'] = _np(c1.grad), _np(r1.grad)
    # (c) decoding
    bc, bb = head.generate_predicted_boxes(B, torch.from_numpy(rois), rcnn_cls.detach(), rcnn_reg.detach())
    out['roi_decoded'] = _np(bb)
    # (f) FC stack, eval mode
    head.eval()
    pooled = torch.from_numpy(rng.normal(0, 1, (B * R, 8, 16)).astype(np.float32))
    flat = pooled.permute(0, 2, 1).contiguous().view(B * R, -1, 2, 2, 2).view(B * R, -1, 1)
    shared = head.shared_fc_layer(flat)
    out['fc_state'] = {k: _np(v) for k, v in head.state_dict().items() if not k.startswith('roi_grid_pool_layer')}
    out['fc_pooled'] = _np(pooled)
    out['fc_cls'] = _np(head.cls_layers(shared).transpose(1, 2).contiguous().squeeze(1))
    out['fc_reg'] = _np(head.reg_layers(shared).transpose(1, 2).contiguous().squeeze(1))
    g, l = head.get_global_grid_points_of_roi(torch.from_numpy(rois), 2)
    out['grid_global'] = _np(g)


def gen_strategies(out):
    """arithmetic of the baseline query strategies, run through the reference's own methods"""
    from pcdet.query_strategies.coreset_sampling import CoresetSampling
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    me = CoresetSampling.__new__(CoresetSampling)               # furthest_first / pairwise only touch their arguments
    X = torch.from_numpy(rng.normal(size=(23, 6, 5)).astype(np.float32))
    X[7] = X[3]                                                 # a duplicate row: ties in the arg-max
    S = torch.from_numpy(rng.normal(size=(4, 6, 5)).astype(np.float32) + 0.5)
    out['cs_X'], out['cs_S'] = _np(X), _np(S)
    out['cs_dist'] = _np(me.pairwise_squared_distances(X, S))
    out['cs_pick'] = np.array([int(i) for i in me.furthest_first(X.clone(), S.clone(), 9)], np.int64)
    # confidence / BALD value of a frame and the selection rule (confidence_sampling.py:44-66)
    logits = [torch.from_numpy(rng.normal(size=(int(k), 3)).astype(np.float32) * 2) for k in (5, 1, 12, 7, 3, 9)]
    vals = [(-(F.softmax(l, dim=1) * F.log_softmax(l, dim=1)).sum(dim=1)).mean() for l in logits]
    sel = dict(sorted({i: v for i, v in enumerate(vals)}.items(), key=lambda item: item[1]))
    out['ent_logits'] = np.concatenate([_np(l) for l in logits])
    out['ent_counts'] = np.array([len(l) for l in logits], np.int64)
    out['ent_vals'] = np.array([float(v) for v in vals], np.float32)
    out['ent_selected'] = np.array(list(sel.keys())[len(sel) - 3:], np.int64)


def gen_badge(out):
    """BADGE embedding at the level the strategy computes it (badge_sampling.py:84-90,146-160): eval pass of the reference's
    AnchorHeadSingle -> rpn_preds -> arg-max over classes = hypothetical labels; then, frame by frame (bs = 1, train mode),
    get_cls_layer_loss(new_data={'box_cls_labels', 'cls_preds'}) (anchor_head_template.py:101-142) -> backward ->
    conv_cls.weight.grad = the embedding row. Inputs are the BEV features the detector would hand the head."""
    from pcdet.models.dense_heads.anchor_head_single import AnchorHeadSingle
    rng = np.random.default_rng(31)
    pc_range = np.array([0, -8, -3, 17.6, 8, 1], np.float32)
    grid = np.array([176, 160, 40], np.int64)
    torch.manual_seed(32)
    head = AnchorHeadSingle(small_head_cfg(), input_channels=24, num_class=3,
                            class_names=['Car', 'Pedestrian', 'Cyclist'], grid_size=grid, point_cloud_range=pc_range,
                            predict_boxes_when_training=True)
    torch.nn.init.normal_(head.conv_cls.weight, std=0.3)          # away from the -log(99) bias-only start: mixed labels
    B = 3
    feats = rng.normal(0, 1, (B, 24, 20, 22)).astype(np.float32)
    head.eval()
    with torch.no_grad():
        dd = head({'spatial_features_2d': torch.from_numpy(feats), 'batch_size': B})
    rpn = dd['rpn_preds']
    labels = torch.argmax(rpn.view(B, -1, head.num_class), -1)
    head.train()
    gt = rand_gt(rng, B, 5, ((0.5, 17), (-7.5, 7.5)))
    embs, losses = [], []
    for b in range(B):
        d1 = head({'spatial_features_2d': torch.from_numpy(feats[b:b + 1]), 'gt_boxes': torch.from_numpy(gt[b:b + 1].copy()),
                   'batch_size': 1})
        new_data = {'box_cls_labels': labels[b].unsqueeze(0), 'cls_preds': d1['rpn_preds']}
        loss = head.get_cls_layer_loss(new_data=new_data)[0]
        head.zero_grad()
        loss.backward()
        embs.append(_np(head.conv_cls.weight.grad).reshape(-1).copy())
        losses.append(float(loss))
    out['badge_state'] = {k: _np(v) for k, v in head.state_dict().items()}
    out['badge_feats'], out['badge_gt'] = feats, gt
    out['badge_rpn_preds'] = _np(rpn)
    out['badge_labels'] = _np(labels).astype(np.int8)
    out['badge_loss'] = np.array(losses, np.float64)
    out['badge_emb'] = np.stack(embs)
    assert len(np.unique(out['badge_labels'])) == 3 and np.abs(out['badge_emb']).max() > 0


def gen_partA2(out):
    """ref_partA2.npz: the reference's PartA2FCHead (partA2_head.py:10-224) end to end — RoI-aware pooling (avg part
    features, max point features), occupied cells of all RoI grids as one sparse tensor, the two SubM conv stacks, dense
    flatten, FC branches, box decoding — in eval mode (running BatchNorm statistics) and in train mode (batch statistics;
    proposal / target assignment bypassed with the same fixed RoIs, which partA2_head.py:165-172 allows through
    batch_dict['rois']). Compiled entry points are answered by the oracle: roiaware_pool3d_cuda.forward -> oracle
    roiaware_pool (points-in-box rule pinned to roiaware_pool3d.cpp), spconv.SubMConv3d / SparseConvTensor.dense ->
    oracle subm_nbr + conv_fwd + dense (spconv itself is not in the reference tree: restated semantics)."""
    oracle = _install_cpu_ops()
    pool_mod = sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda']

    def pool_forward(rois, pts, feat, argmax, pts_idx, pooled, method):
        o = tuple(int(v) for v in pooled.shape[1:4])
        p, a, i = oracle.roiaware_pool(rois.numpy(), pts.numpy(), feat.detach().numpy(), o, int(pts_idx.shape[-1]), int(method))
        pooled.copy_(torch.from_numpy(p)); argmax.copy_(torch.from_numpy(a)); pts_idx.copy_(torch.from_numpy(i))
    pool_mod.forward = pool_forward

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = [int(v) for v in spatial_shape], int(batch_size)

        def replace_feature(self, f):
            return SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)

        def dense(self):
            return torch.from_numpy(oracle.dense(self.features.detach().numpy(), self.indices.numpy(), self.batch_size,
                                                 self.spatial_shape))

    class SubMConv3d(torch.nn.Module):
        def __init__(self, cin, cout, k, bias=True, indice_key=None, **kw):
            super().__init__()
            assert k == 3 and not bias
            self.weight = torch.nn.Parameter(torch.zeros(cout, 3, 3, 3, cin))      # spconv 2.x layout

        def forward(self, x):
            nbr = oracle.subm_nbr(x.indices.numpy(), x.spatial_shape, [3, 3, 3])
            w = self.weight.detach().reshape(self.weight.shape[0], 27, -1).permute(1, 2, 0).contiguous().numpy()
            return x.replace_feature(torch.from_numpy(oracle.conv_fwd(x.features.detach().numpy(), w, nbr)))

    class SparseSequential(torch.nn.Sequential):
        def forward(self, x):
            for m in self:
                x = m(x) if isinstance(m, (SubMConv3d, SparseSequential)) else x.replace_feature(m(x.features))
            return x

    spp = sys.modules['spconv.pytorch']
    spp.SparseConvTensor, spp.SubMConv3d, spp.SparseSequential = SparseConvTensor, SubMConv3d, SparseSequential
    from pcdet.config import cfg as ref_cfg
    ref_cfg.CLASS_NAMES = ['Car', 'Pedestrian', 'Cyclist']
    from pcdet.models.roi_heads.partA2_head import PartA2FCHead
    head = PartA2FCHead(input_channels=32, model_cfg=parta2_cfg(), num_class=1)
    head.load_state_dict(seeded_state(head, 43))
    inp = parta2_inputs()

    def batch():
        return {k: (torch.from_numpy(v.copy()) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
    head.eval()
    with torch.no_grad():
        bd = head(batch())
        pooled_part, pooled_rpn = head.roiaware_pool(batch())
    out['pa2_keys'] = np.array(sorted(head.state_dict().keys()))
    out['pa2_pooled_part'], out['pa2_pooled_rpn'] = _np(pooled_part), _np(pooled_rpn)
    out['pa2_eval_cls'], out['pa2_eval_box'] = _np(bd['batch_cls_preds']), _np(bd['batch_box_preds'])
    head.train()
    head.assign_targets = lambda bdict: {'rois': bdict['rois'], 'roi_labels': bdict['roi_labels']}
    with torch.no_grad():
        head(batch())
    out['pa2_train_cls'], out['pa2_train_reg'] = _np(head.forward_ret_dict['rcnn_cls']), _np(head.forward_ret_dict['rcnn_reg'])
    cells = int((out['pa2_pooled_part'].sum(-1) != 0).sum())
    assert cells > 200 and np.abs(out['pa2_eval_cls']).max() > 1e-3 and np.isfinite(out['pa2_train_reg']).all(), cells
    print('  occupied cells', cells, 'of', out['pa2_pooled_part'][..., 0].size)


def _install_pointnet2_ops(oracle):
    """answer pointnet2_stack_cuda's entry points (ball query, grouping + its gradient, farthest point sampling) with the
    oracle's loop-for-loop restatement of the reference's .cu kernels (oracle/pointnet2_oracle.c); the reference's autograd
    Functions, QueryAndGroup, StackSAModuleMSG and every caller above them run unmodified"""
    m = sys.modules['pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda']

    def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_cnt, xyz, xyz_cnt, idx):
        idx.copy_(torch.from_numpy(oracle.ball_query(float(radius), int(nsample), xyz.detach().numpy(), xyz_cnt.numpy(),
                                                     new_xyz.detach().numpy(), new_cnt.numpy())))

    def group_points_wrapper(B, M, C, nsample, feat, feat_cnt, idx, idx_cnt, out):
        out.copy_(torch.from_numpy(oracle.group_points(feat.detach().numpy(), feat_cnt.numpy(), idx.numpy(), idx_cnt.numpy())))

    def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_cnt, feat_cnt, grad_feat):
        grad_feat.copy_(torch.from_numpy(oracle.group_points_grad(grad_out.numpy(), idx.numpy(), idx_cnt.numpy(),
                                                                  feat_cnt.numpy(), int(N))))

    def farthest_point_sampling_wrapper(B, N, npoint, xyz, temp, out):
        out.copy_(torch.from_numpy(oracle.fps(xyz.numpy(), int(npoint))))
    m.ball_query_wrapper, m.group_points_wrapper = ball_query_wrapper, group_points_wrapper
    m.group_points_grad_wrapper, m.farthest_point_sampling_wrapper = group_points_grad_wrapper, farthest_point_sampling_wrapper
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor


def _ref_pfe_cfg():
    import yaml
    y = yaml.safe_load(open(os.path.join(REF, 'tools/cfgs/active-kitti_models/pv_rcnn_active_crb.yaml')))
    pfe, roi = EasyDict(y['MODEL']['PFE']), EasyDict(y['MODEL']['ROI_HEAD'])
    pfe.NUM_KEYPOINTS = PP_KEYPOINTS
    return pfe, roi


def gen_point_path(out):
    """ref_point_path.npz: the PV-RCNN point path as the reference's own classes compose it —
    VoxelSetAbstraction.forward (voxel_set_abstraction.py:284-411: FPS keypoints, bilinear BEV lookup, raw-point SA, the four
    voxel-centre SAs in FEATURES_SOURCE order, concat, fusion Linear+BN+ReLU), StackSAModuleMSG.forward
    (pointnet2_modules.py:78-112) underneath each source, and PVRCNNHead.roi_grid_pool (pvrcnn_head.py:68-114) on the result
    — with the PFE / ROI_GRID_POOL sections of the reference's own pv_rcnn_active_crb.yaml (only NUM_KEYPOINTS reduced),
    in eval mode (running statistics) and train mode (batch statistics + a backward pass through GroupingOperation.backward).
    pointnet2_stack_cuda's entry points are answered by the oracle."""
    oracle = _install_cpu_ops()
    _install_pointnet2_ops(oracle)
    from pcdet.config import cfg as ref_cfg
    ref_cfg.CLASS_NAMES = ['Car', 'Pedestrian', 'Cyclist']
    from pcdet.models.backbones_3d.pfe.voxel_set_abstraction import VoxelSetAbstraction
    from pcdet.models.roi_heads.pvrcnn_head import PVRCNNHead
    pfe_cfg, roi_cfg = _ref_pfe_cfg()
    inp = point_path_inputs()
    c_bev = inp['bev'].shape[1]
    vsa = VoxelSetAbstraction(pfe_cfg, voxel_size=PP_VOXEL, point_cloud_range=PP_PCR, num_bev_features=c_bev,
                              num_rawpoint_features=4)
    vsa.load_state_dict(seeded_state(vsa, 53))
    head = PVRCNNHead(input_channels=vsa.num_point_features, model_cfg=roi_cfg, num_class=1)
    head.load_state_dict(seeded_state(head, 57))
    out['pp_vsa_keys'] = np.array(sorted(vsa.state_dict().keys()))
    out['pp_head_pool_keys'] = np.array(sorted(k for k in head.state_dict() if k.startswith('roi_grid_pool_layer')))

    class Level:
        def __init__(self, c, f):
            self.indices, self.features = torch.from_numpy(c.copy()), torch.from_numpy(f.copy())

    def batch():
        return {'batch_size': inp['batch_size'], 'points': torch.from_numpy(inp['points'].copy()),
                'multi_scale_3d_features': {k: Level(*v) for k, v in inp['levels'].items()},
                'spatial_features': torch.from_numpy(inp['bev'].copy()), 'spatial_features_stride': 8,
                'rois': torch.from_numpy(inp['rois'].copy())}

    for mode in ('eval', 'train'):
        vsa.train(mode == 'train'); head.train(mode == 'train')
        bd = batch()
        with torch.set_grad_enabled(mode == 'train'):
            bd = vsa(bd)
            bd['point_cls_scores'] = torch.from_numpy(inp['scores'].copy())
            pooled = head.roi_grid_pool(bd)
        out['pp_%s_before_fusion' % mode] = _np(bd['point_features_before_fusion'])
        out['pp_%s_point_features' % mode] = _np(bd['point_features'])
        out['pp_%s_pooled' % mode] = _np(pooled)
        if mode == 'eval':
            out['pp_point_coords'] = _np(bd['point_coords'])
        else:
            (pooled.square().sum() + bd['point_features'].square().sum()).backward()
            for n in ('SA_rawpoints.mlps.0.0.weight', 'SA_layers.0.mlps.1.3.weight', 'SA_layers.3.mlps.1.0.weight',
                      'vsa_point_feature_fusion.0.weight'):
                out['pp_grad/' + n] = _np(dict(vsa.named_parameters())[n].grad)
            out['pp_grad/roi_grid_pool_layer.mlps.0.0.weight'] = _np(head.roi_grid_pool_layer.mlps[0][0].weight.grad)
    # the running statistics after the train pass are part of the contract too (momentum update of every BN layer)
    out['pp_running_mean_after'] = _np(vsa.vsa_point_feature_fusion[1].running_mean).copy()    # (shares memory otherwise)
    # StackSAModuleMSG.forward on its own with ragged counts on both sides (first frame 40 queries, second 7)
    vsa.load_state_dict(seeded_state(vsa, 53))          # the train pass above advanced the running statistics
    sa = vsa.SA_layers[1]
    sa.eval()
    c, f = inp['levels']['x_conv2']
    from pcdet.utils import common_utils
    xyz = common_utils.get_voxel_centers(torch.from_numpy(c[:, 1:4].copy()), 2, PP_VOXEL, PP_PCR)
    cnt = torch.from_numpy(np.bincount(c[:, 0], minlength=2).astype(np.int32))
    kp = out['pp_point_coords']
    q = np.concatenate([kp[kp[:, 0] == 0][:40, 1:4], kp[kp[:, 0] == 1][:7, 1:4]]).astype(np.float32)
    with torch.no_grad():
        _, y = sa(xyz=xyz.contiguous(), xyz_batch_cnt=cnt, new_xyz=torch.from_numpy(q),
                  new_xyz_batch_cnt=torch.tensor([40, 7], dtype=torch.int32), features=torch.from_numpy(f.copy()))
    out['pp_sa_queries'], out['pp_sa_out'] = q, _np(y)
    print('  keypoints', kp.shape, 'before_fusion', out['pp_eval_before_fusion'].shape, 'pooled', out['pp_eval_pooled'].shape)
    assert out['pp_eval_before_fusion'].shape[1] == c_bev + 32 * 2 + 64 + 128 + 128
    assert np.isfinite(out['pp_train_pooled']).all() and np.abs(out['pp_grad/SA_rawpoints.mlps.0.0.weight']).max() > 0


def gen_data_processor(out):
    """DataProcessor.mask_points_and_boxes_outside_range through the reference's own class (train mode, shuffle off)"""
    from pcdet.datasets.processor.data_processor import DataProcessor
    rng = np.random.default_rng(21)
    pcr = np.array([0, -40, -3, 70.4, 40, 1], dtype=np.float32)
    cfgs = [EasyDict({'NAME': 'mask_points_and_boxes_outside_range', 'REMOVE_OUTSIDE_BOXES': True}),
            EasyDict({'NAME': 'shuffle_points', 'SHUFFLE_ENABLED': EasyDict({'train': False, 'test': False})}),
            EasyDict({'NAME': 'transform_points_to_voxels_placeholder', 'VOXEL_SIZE': [0.05, 0.05, 0.1]})]
    dp = DataProcessor(cfgs, point_cloud_range=pcr, training=True, num_point_features=4)
    pts = rng.uniform([-8, -48, -4, 0], [78, 48, 2, 1], size=(3000, 4)).astype(np.float32)
    gt = np.concatenate([rng.uniform([-5, -45, -2], [76, 45, 0], size=(14, 3)), rng.uniform(1.0, 4.5, size=(14, 3)),
                         rng.uniform(-3.1, 3.1, size=(14, 1)), rng.integers(1, 4, size=(14, 1))], 1).astype(np.float32)
    d = dp.forward({'points': pts.copy(), 'gt_boxes': gt.copy(), 'use_lead_xyz': True})
    out['dp_points_in'], out['dp_gt_in'] = pts, gt
    out['dp_points_out'], out['dp_gt_out'] = d['points'], d['gt_boxes']
    out['dp_grid_size'] = dp.grid_size


def save(name, d):
    flat = {}
    for k, v in d.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                flat[k + '/' + k2] = v2
        else:
            flat[k] = v
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **flat)
    print(name, '%.1f KB' % (os.path.getsize(path) / 1024))


# ref_iou3d.npz is produced from oracle/_ref (the reference's iou3d_cpu.cpp compiled by oracle/build_ref.sh):
#   python - <<'PY'
#   import numpy as np, oracle; from boxes_synth import detection_boxes
#   rng = np.random.default_rng(123); a,_ = detection_boxes(rng,160); b,_ = detection_boxes(rng,120)
#   b[:40] = a[:40] + rng.normal(0,0.15,(40,7)).astype(np.float32)
#   np.savez_compressed('tests/golden/ref_iou3d.npz', a=a, b=b, iou=oracle.ref_boxes_iou_bev(a,b))
#   PY


def gen_points_in_boxes_ref():
    """ref_points_in_boxes.npz: the reference's own points_in_boxes_cpu (roiaware_pool3d.cpp:144-167, compiled from
    /root/reference by oracle/build_ref.sh) on boxes with points sampled inside, on the faces (+- the two margins) and
    around them"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    sys.path.insert(0, os.path.dirname(OUT))
    import oracle
    from boxes_synth import detection_boxes
    rng = np.random.default_rng(77)
    boxes, _ = detection_boxes(rng, 40)
    pts = []
    for b in boxes:
        c, s = np.cos(b[6]), np.sin(b[6])
        loc = rng.uniform(-0.75, 0.75, (60, 3)) * b[3:6]
        # face samples: |local x| or |local y| = half extent + {-2e-2, -5e-3, -1e-5, 0, 5e-6, 5e-3, 2e-2}; |dz| = h/2 + ...
        for k, eps in enumerate((-2e-2, -5e-3, -1e-5, 0.0, 5e-6, 5e-3, 2e-2)):
            loc[k, 0] = np.sign(loc[k, 0] + 1e-9) * (b[3] / 2 + eps)
            loc[7 + k, 1] = np.sign(loc[7 + k, 1] + 1e-9) * (b[4] / 2 + eps)
            loc[14 + k, 2] = np.sign(loc[14 + k, 2] + 1e-9) * (b[5] / 2 + eps)
        w = np.stack([loc[:, 0] * c - loc[:, 1] * s + b[0], loc[:, 0] * s + loc[:, 1] * c + b[1], loc[:, 2] + b[2]], 1)
        pts.append(w)
    pts = np.concatenate(pts).astype(np.float32)
    d = {'boxes': boxes, 'points': pts, 'member': oracle.ref_points_in_boxes_cpu(boxes, pts).astype(np.int8)}
    save('ref_points_in_boxes.npz', d)


def _install_cpu_ops():
    """the reference's Python op wrappers stay in play; only the three compiled entry points they call are answered by the
    oracle (pinned to the reference's own CPU sources where those exist: rotated overlap -> iou3d_cpu.cpp, point-in-box
    -> roiaware_pool3d.cpp)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    sys.path.insert(0, os.path.dirname(OUT))
    import oracle
    nms_mod = sys.modules['pcdet.ops.iou3d_nms.iou3d_nms_cuda']
    pool_mod = sys.modules['pcdet.ops.roiaware_pool3d.roiaware_pool3d_cuda']

    def nms_gpu(boxes, keep, thresh):
        k = oracle.nms(boxes.numpy(), float(thresh), rotated=True)
        keep[:len(k)] = torch.from_numpy(k.astype(np.int64))
        return len(k)

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(oracle.boxes_pairwise(a.numpy(), b.numpy(), 0)))
        return 1

    def points_in_boxes_gpu(boxes, points, idx):
        idx.copy_(torch.from_numpy(oracle.points_in_boxes(points.numpy(), boxes.numpy())))
        return 1
    nms_mod.nms_gpu, nms_mod.boxes_overlap_bev_gpu = nms_gpu, boxes_overlap_bev_gpu
    pool_mod.points_in_boxes_gpu = points_in_boxes_gpu
    torch.cuda.FloatTensor = torch.FloatTensor
    return oracle


def post_processing_inputs(seed=31):
    """3 ragged frames: clustered points inside gt boxes + clutter, 128 RoIs around the objects, PV-RCNN-style head outputs"""
    rng = np.random.default_rng(seed)
    sizes = {1: (3.9, 1.6, 1.56), 2: (0.8, 0.6, 1.73), 3: (1.76, 0.6, 1.73)}
    B, G, R = 3, 10, 128
    gt = np.zeros((B, G, 8), np.float32)
    pts, rois = [], np.zeros((B, R, 7), np.float32)
    n_gt = [10, 7, 8]
    for b in range(B):
        for g in range(n_gt[b]):
            c = int(rng.integers(1, 4)) if b != 1 else int(rng.integers(1, 3))          # frame 1 has no Cyclist
            if b == 2:
                c = 2 if g < 2 else int(rng.choice([1, 3]))      # frame 2: two Pedestrian boxes, both without points
            s = np.array(sizes[c]) * rng.uniform(0.9, 1.1, 3)
            gt[b, g] = [rng.uniform(5, 60), rng.uniform(-30, 30), rng.uniform(-1.2, -0.6), *s, rng.uniform(-np.pi, np.pi), c]
        p = [np.stack([rng.uniform(0, 70, 1500), rng.uniform(-40, 40, 1500), rng.uniform(-3, 1, 1500)], 1)]
        for g in range(n_gt[b]):
            if b == 2 and gt[b, g, 7] == 2:
                continue                                     # frame 2: Pedestrian boxes own no point (NaN mean -> 0)
            k = int(rng.integers(1, 60))
            loc = rng.uniform(-0.5, 0.5, (k, 3)) * gt[b, g, 3:6]
            ca, sa = np.cos(gt[b, g, 6]), np.sin(gt[b, g, 6])
            p.append(np.stack([loc[:, 0] * ca - loc[:, 1] * sa + gt[b, g, 0], loc[:, 0] * sa + loc[:, 1] * ca + gt[b, g, 1],
                               loc[:, 2] + gt[b, g, 2]], 1))
        p = np.concatenate(p).astype(np.float32)
        if b == 2:                                           # keep the clutter away from the point-free boxes
            far = np.ones(len(p), bool)
            for g in range(n_gt[b]):
                if gt[b, g, 7] == 2:
                    far &= np.linalg.norm(p[:, :2] - gt[b, g, :2], axis=1) > 2.0
            p = p[far]
        p = p[rng.permutation(len(p))]
        pts.append(np.concatenate([np.full((len(p), 1), b, np.float32), p, rng.uniform(0, 1, (len(p), 1)).astype(np.float32)], 1))
        k = rng.integers(0, n_gt[b], R)
        rois[b] = gt[b, k, :7] + rng.normal(0, 0.3, (R, 7)).astype(np.float32) * np.array([1, 1, .2, .1, .1, .1, .3], np.float32)
        rois[b, 100:, :2] += rng.uniform(8, 20, (28, 2)).astype(np.float32)            # some boxes off the objects
    points = np.concatenate(pts).astype(np.float32)
    box_preds = rois + rng.normal(0, 0.05, rois.shape).astype(np.float32)
    d = {
        'points': points, 'gt_boxes': gt, 'rois': rois, 'batch_box_preds': box_preds,
        'batch_cls_preds': rng.normal(0.3, 2.0, (B, R, 1)).astype(np.float32),
        'roi_labels': rng.integers(1, 4, (B, R)).astype(np.int64),
        'full_cls_scores': rng.normal(0, 1.5, (B, R, 3)).astype(np.float32),
        'rcnn_cls': rng.normal(0, 1.5, (5, B * R, 1)).astype(np.float32),
        'rcnn_reg': rng.normal(0, 0.4, (5, B * R, 7)).astype(np.float32),
        'rpn_preds': rng.normal(0, 1, (B, 2, 3, 18)).astype(np.float32),
    }
    return d


def gen_post_processing(out):
    """ref_post_processing.npz + ref_selected_frames_epoch_7_rank_0.pkl: the reference's CRB-patched
    Detector3DTemplate.post_processing (detector3d_template.py:186-409) on a 3-frame batch -> its 15 record keys per frame,
    the recall dict, and the pickle its Strategy.save_points / save_active_labels (strategy.py:28-38,66-75) write for two
    selected frames"""
    import shutil
    import tempfile
    import types
    _install_cpu_ops()
    from pcdet.models.detectors.detector3d_template import Detector3DTemplate
    from pcdet.query_strategies.strategy import Strategy
    inp = post_processing_inputs()
    names = ['Car', 'Pedestrian', 'Cyclist']
    model_cfg = EasyDict({'POST_PROCESSING': POST_CFG,
                          'DENSE_HEAD': {'ANCHOR_GENERATOR_CONFIG': [{'class_name': n} for n in names]}})
    fake = types.SimpleNamespace(model_cfg=model_cfg, num_class=3,
                                 generate_recall_record=Detector3DTemplate.generate_recall_record)
    bd = {k: torch.from_numpy(v.copy()) for k, v in inp.items()}
    bd.update({'batch_size': 3, 'cls_preds_normalized': False, 'has_class_labels': True})
    with torch.no_grad():
        pred_dicts, recall = Detector3DTemplate.post_processing(fake, bd)
    for k, v in inp.items():
        out['in_' + k] = v
    f = lambda v: np.float64(v) if not torch.is_tensor(v) else _np(v).astype(np.float64)
    for b, d in enumerate(pred_dicts):
        for k in ('pred_boxes', 'pred_scores', 'pred_labels', 'pred_logits', 'pred_box_unique_density', 'batch_rcnn_cls',
                  'batch_rcnn_reg', 'confidence'):
            out['f%d_%s' % (b, k)] = _np(d[k])
        for k in ('num_bbox', 'mean_points', 'median_points', 'variance_points'):
            out['f%d_%s' % (b, k)] = np.array([f(d[k][n]) for n in names])
            out['f%d_%s_is_tensor' % (b, k)] = np.array([torch.is_tensor(d[k][n]) for n in names])
        assert d['loss_predictions'] is None and d['embeddings'] is None
    out['recall_keys'] = np.array(sorted(recall.keys()))
    out['recall_vals'] = np.array([recall[k] for k in sorted(recall.keys())], np.int64)
    # the reference's bookkeeping + pickle
    tmp = tempfile.mkdtemp()
    st = Strategy.__new__(Strategy)
    st.active_label_dir, st.rank = tmp, 0
    st.bbox_records = {}
    st.point_measures = ['mean', 'median', 'variance']
    for met in st.point_measures:
        setattr(st, '{}_point_records'.format(met), {})
    ids = ['000010', '000011', '000012']
    for b, d in enumerate(pred_dicts):
        st.save_points(ids[b], d)
    st.save_active_labels(selected_frames=['000012', '000010'], cur_epoch=7)
    shutil.copy(os.path.join(tmp, 'selected_frames_epoch_7_rank_0.pkl'),
                os.path.join(OUT, 'ref_selected_frames_epoch_7_rank_0.pkl'))
    shutil.rmtree(tmp)
    out['frame_ids'] = np.array(ids)


def gen_glue(out):
    """ref_glue.npz: host/device glue rows of SURVEY §8 that had no reference pin — DatasetTemplate.collate_batch
    (dataset.py:160-229), the eval DistributedSampler (datasets/__init__.py:26-46), bilinear_interpolate_torch +
    VoxelSetAbstraction.interpolate_from_bev_features (voxel_set_abstraction.py:11-42,176-204), PointHeadSimple forward /
    assign_stack_targets / loss (point_head_simple.py:58-91, point_head_template.py:49-170)"""
    import types
    oracle = _install_cpu_ops()
    from pcdet.datasets import DistributedSampler
    from pcdet.datasets.dataset import DatasetTemplate
    from pcdet.models.backbones_3d.pfe import voxel_set_abstraction as vsa
    from pcdet.models.dense_heads.point_head_simple import PointHeadSimple
    rng = np.random.default_rng(41)
    # ---- collate_batch: 3 ragged frames in the reference loader's layout
    frames = []
    for i, (n, m, g) in enumerate(((50, 17, 4), (31, 9, 0), (44, 20, 7))):
        frames.append({'points': rng.normal(size=(n, 4)).astype(np.float32),
                       'voxels': rng.normal(size=(m, 5, 4)).astype(np.float32),
                       'voxel_coords': rng.integers(0, 40, (m, 3)).astype(np.int32),
                       'voxel_num_points': rng.integers(1, 6, m).astype(np.int32),
                       'gt_boxes': rng.normal(size=(g, 8)).astype(np.float32), 'frame_id': '%06d' % (7 * i + 3),
                       'use_lead_xyz': True})
    col = DatasetTemplate.collate_batch([dict(f) for f in frames])
    for i, f in enumerate(frames):
        for k in ('points', 'voxels', 'voxel_coords', 'voxel_num_points', 'gt_boxes'):
            out['col_in%d_%s' % (i, k)] = f[k]
    out['col_in_frame_id'] = np.array([f['frame_id'] for f in frames])
    for k, v in col.items():
        out['col_out_' + k] = np.asarray(v)
    # ---- eval sampler: every (n, world, rank)
    for n, world in ((10, 4), (3000, 8), (7, 2), (5, 8), (16, 1)):
        out['sampler_%d_%d' % (n, world)] = np.stack([
            np.array(list(DistributedSampler(list(range(n)), world, r, shuffle=False)), np.int64) for r in range(world)])
    # ---- bilinear lookup incl. positions outside the map (clamped taps)
    im = rng.normal(size=(25, 22, 6)).astype(np.float32)
    x = rng.uniform(-1.5, 23.5, 300).astype(np.float32)
    y = rng.uniform(-1.5, 26.5, 300).astype(np.float32)
    x[:5], y[:5] = [0, 21, 3, 21.0, 10.5], [0, 24, 24.0, 0.5, 7]
    out['bil_im'], out['bil_x'], out['bil_y'] = im, x, y
    out['bil_out'] = _np(vsa.bilinear_interpolate_torch(torch.from_numpy(im), torch.from_numpy(x), torch.from_numpy(y)))
    fake = types.SimpleNamespace(voxel_size=[0.05, 0.05, 0.1], point_cloud_range=np.array([0, -40, -3, 70.4, 40, 1], np.float32))
    B, K = 3, 40
    kp = np.concatenate([np.repeat(np.arange(B), K)[:, None], rng.uniform(0, 8.8, (B * K, 1)), rng.uniform(-40, -30, (B * K, 1)),
                         rng.uniform(-3, 1, (B * K, 1))], 1).astype(np.float32)
    bev = rng.normal(size=(B, 6, 25, 22)).astype(np.float32)
    out['bev_kp'], out['bev_map'] = kp, bev
    out['bev_out'] = _np(vsa.VoxelSetAbstraction.interpolate_from_bev_features(fake, torch.from_numpy(kp), torch.from_numpy(bev), B, 8))
    # ---- point head
    cfg = EasyDict({'NAME': 'PointHeadSimple', 'CLS_FC': [16, 16], 'CLASS_AGNOSTIC': True,
                    'USE_POINT_FEATURES_BEFORE_FUSION': True, 'NUM_KEYPOINTS': 64,
                    'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2]},
                    'LOSS_CONFIG': {'LOSS_REG': 'smooth-l1', 'LOSS_WEIGHTS': {'point_cls_weight': 1.0}}})
    torch.manual_seed(9)
    head = PointHeadSimple(num_class=1, input_channels=12, model_cfg=cfg)
    head.train()
    K = 64
    gt = rand_gt(rng, B, 6, ((2, 30), (-12, 12)))
    pc = []
    for b in range(B):
        p = np.stack([rng.uniform(0, 32, K), rng.uniform(-14, 14, K), rng.uniform(-2, 0.5, K)], 1)
        for j in range(K // 2):                                        # half the keypoints on / just around the objects
            g = gt[b, j % 6]
            if g[7] == 0:
                continue
            loc = rng.uniform(-0.58, 0.58, 3) * g[3:6]
            ca, sa = np.cos(g[6]), np.sin(g[6])
            p[j] = [loc[0] * ca - loc[1] * sa + g[0], loc[0] * sa + loc[1] * ca + g[1], loc[2] + g[2]]
        pc.append(np.concatenate([np.full((K, 1), b), p], 1))
    pc = np.concatenate(pc).astype(np.float32)
    feats = torch.from_numpy(rng.normal(size=(B * K, 12)).astype(np.float32)).requires_grad_(True)
    bd = head({'point_features_before_fusion': feats, 'point_features': feats, 'point_coords': torch.from_numpy(pc),
               'gt_boxes': torch.from_numpy(gt.copy()), 'batch_size': B})
    loss, tb = head.get_loss()
    loss.backward()
    out['ph_state'] = {k: _np(v) for k, v in head.state_dict().items()}
    out['ph_feats'], out['ph_coords'], out['ph_gt'] = _np(feats), pc, gt
    out['ph_labels'] = _np(head.forward_ret_dict['point_cls_labels'])
    out['ph_scores'] = _np(bd['point_cls_scores'])
    out['ph_loss'] = np.array([float(loss), float(tb['point_loss_cls']), float(tb['point_pos_num'])])
    out['ph_feats_grad'] = _np(feats.grad)
    assert (out['ph_labels'] == -1).sum() > 3 and (out['ph_labels'] == 1).sum() > 10


# ---------------------------------------------------------------------------------------------------------------------
# configs[2] at the detector level: the reference's own PVRCNN (pcdet/models/detectors/pv_rcnn.py:9-43) — every module of
# build_networks() and get_training_loss() — on the CPU; the compiled ops and spconv are answered by the oracle.
# ---------------------------------------------------------------------------------------------------------------------
# kind -> (reference yaml, point cloud range, voxel size, point features, points per frame, voxel cap)
# gradients stored (one per module of the detector) and the slice of each that is kept (the whole tensors would be 30 MB)


def _install_spconv_oracle(oracle):
    """spconv.pytorch as the reference's backbone / map_to_bev use it (spconv_backbone.py:8-157, height_compression.py:20-24),
    answered by the oracle's restated semantics (sparse_conv_oracle.c: rulebooks, forward, input and weight gradient) wrapped
    as autograd Functions; strided outputs in ascending linear (b,z,y,x) order. spconv itself is not in the reference tree."""
    from oracle.second_cpu import _OracleConv

    class SparseConvTensor:
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices = features, indices
            self.spatial_shape, self.batch_size = [int(v) for v in spatial_shape], int(batch_size)
            self.rulebooks = {}

        def replace_feature(self, f):
            t = SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)
            t.rulebooks = self.rulebooks
            return t

        def dense(self):
            c = self.indices.long()
            d, h, w = self.spatial_shape
            out = torch.zeros(self.batch_size, d, h, w, self.features.shape[1])
            out = out.index_put((c[:, 0], c[:, 1], c[:, 2], c[:, 3]), self.features)
            return out.permute(0, 4, 1, 2, 3).contiguous()

    class SparseConvolution(torch.nn.Module):
        def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, indice_key=None, subm=False):
            super().__init__()
            assert not bias
            t3 = lambda v: [int(v)] * 3 if np.isscalar(v) else [int(x) for x in v]
            self.kernel_size, self.stride, self.padding, self.subm, self.indice_key = t3(k), t3(stride), t3(padding), subm, indice_key
            self.weight = torch.nn.Parameter(torch.zeros(cout, *self.kernel_size, cin))      # spconv 2.x layout

        def forward(self, x):
            coords = x.indices.numpy()
            K = int(np.prod(self.kernel_size))
            if self.subm:
                if self.indice_key not in x.rulebooks:
                    x.rulebooks[self.indice_key] = oracle.subm_nbr(coords, x.spatial_shape, self.kernel_size)
                nbr, out = x.rulebooks[self.indice_key], x
            else:
                oc, oshape = oracle.spconv_out(coords, x.spatial_shape, self.kernel_size, self.stride, self.padding)
                nbr = oracle.spconv_nbr(coords, x.spatial_shape, oc, self.kernel_size, self.stride, self.padding)
                out = SparseConvTensor(None, torch.from_numpy(oc), oshape, x.batch_size)
            w = self.weight.reshape(self.weight.shape[0], K, -1).permute(1, 2, 0).contiguous()           # (K, Cin, Cout)
            return out.replace_feature(_OracleConv.apply(x.features, w, nbr, len(coords)))

    class SubMConv3d(SparseConvolution):
        def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, indice_key=None, **kw):
            super().__init__(cin, cout, k, 1, padding, bias, indice_key, subm=True)

    class SparseConv3d(SparseConvolution):
        def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, indice_key=None, **kw):
            super().__init__(cin, cout, k, stride, padding, bias, indice_key, subm=False)

    class SparseSequential(torch.nn.Sequential):
        def forward(self, x):
            for m in self:
                x = m(x) if isinstance(m, (SparseConvolution, SparseSequential)) else x.replace_feature(m(x.features))
            return x

    spp = sys.modules['spconv.pytorch']
    spp.SparseConvTensor, spp.SubMConv3d, spp.SparseConv3d = SparseConvTensor, SubMConv3d, SparseConv3d
    spp.SparseSequential, spp.SparseModule = SparseSequential, torch.nn.Module
    spp.conv.SparseConvolution = SparseConvolution


def pvrcnn_detector_inputs(kind='kitti'):
    """two synthetic frames of tests/synth.py (the generator the GPU tests use): KITTI-shaped, or Waymo-shaped (360 degrees,
    5 point features)"""
    import importlib.util                       # by path: the name `pcdet` is the reference's package in this process
    spec = importlib.util.spec_from_file_location(
        'crb_synthetic', os.path.join(os.path.dirname(os.path.dirname(OUT)), 'crb-active-3ddet_amd', 'pcdet', 'datasets', 'synthetic.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kitti_batch(PV_FIRST_FRAME, 2, PV_KINDS[kind][4], waymo=(kind == 'waymo'))


def pvrcnn_model_cfg(kind='kitti'):
    """MODEL section of the reference's pv_rcnn_active_crb.yaml (KITTI or Waymo) with the two changes the golden needs: 256
    keypoints (CPU time) and no dropout in the RoI head (DP_RATIO 0: train-mode dropout draws are not reproducible across
    implementations)"""
    import yaml
    y = yaml.safe_load(open(os.path.join(REF, PV_KINDS[kind][0])))
    m = EasyDict(y['MODEL'])
    m.PFE.NUM_KEYPOINTS = PV_KEYPOINTS
    m.ROI_HEAD.DP_RATIO = 0.0
    return m, y['CLASS_NAMES']


def gen_pvrcnn_detector(out, kind='kitti'):
    """ref_pvrcnn_detector.npz: one training step of the reference's PVRCNN on two frames — loss, every tb_dict entry, the
    second-stage outputs and eight parameter gradients (one per module of the detector). The RoI sampler's indices
    (proposal_target_layer.py:116-160: np.random / CPU torch.randint draws) are recorded and stored: the test injects them."""
    oracle = _install_cpu_ops()
    _install_pointnet2_ops(oracle)
    _install_spconv_oracle(oracle)
    from pcdet.config import cfg as ref_cfg
    model_cfg, class_names = pvrcnn_model_cfg(kind)
    _, pcr_l, vs_l, n_feat, _, max_vox = PV_KINDS[kind]
    ref_cfg.CLASS_NAMES = class_names
    ref_cfg.MODEL = model_cfg
    from pcdet.models import build_network
    from pcdet.models.roi_heads.target_assigner.proposal_target_layer import ProposalTargetLayer
    pcr, vs = np.array(pcr_l, np.float32), np.array(vs_l, np.float32)
    grid = np.round((pcr[3:6] - pcr[0:3]) / vs).astype(np.int64)

    class Dataset:
        pass
    ds = Dataset()
    ds.class_names, ds.grid_size, ds.point_cloud_range, ds.voxel_size = class_names, grid, pcr, list(vs)
    ds.depth_downsample_factor = None
    ds.point_feature_encoder = EasyDict(num_point_features=n_feat)
    torch.manual_seed(0)
    model = build_network(model_cfg=model_cfg, num_class=3, dataset=ds)
    model.load_state_dict(pv_seeded_state(model))
    model.train()
    out['pv_keys'] = np.array(sorted(model.state_dict().keys()))

    pts, off, gt0 = pvrcnn_detector_inputs(kind)
    B = len(off) - 1
    voxels, coords, npts, _ = oracle.voxelize_batch(pts, off, pcr[:3], vs, grid, max_vox, 5)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))

    def make_batch(gt):
        return {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)), 'voxels': torch.from_numpy(voxels.copy()),
                'voxel_coords': torch.from_numpy(coords.astype(np.float32)), 'voxel_num_points': torch.from_numpy(npts.astype(np.float32)),
                'gt_boxes': torch.from_numpy(gt.copy()), 'batch_size': B,
                'frame_id': np.array(['%06d' % (PV_FIRST_FRAME + i) for i in range(B)])}
    # pass 1 (no gradients): the proposals of the randomly initialised first stage. Ground truth is an INPUT: boxes are placed
    # on some of those proposals (slightly moved / scaled / turned, class = the proposal's label) so that the second stage has
    # foreground RoIs and its regression / corner losses are exercised; the forward pass up to the proposals does not read them.
    captured = {}
    orig_pl = model.roi_head.proposal_layer

    def capture(bd, nms_config):
        bd = orig_pl(bd, nms_config=nms_config)
        captured['rois'], captured['labels'] = bd['rois'].detach().clone(), bd['roi_labels'].detach().clone()
        return bd
    model.roi_head.proposal_layer = capture
    np.random.seed(5); torch.manual_seed(5)
    with torch.no_grad():
        model(make_batch(gt0))
    model.roi_head.proposal_layer = orig_pl
    rng = np.random.default_rng(77)
    gt = np.zeros((B, 10, 8), np.float32)
    for b in range(B):
        k = 0
        for r, lab in zip(captured['rois'][b].numpy(), captured['labels'][b].numpy()):
            ok = (r[3:6] > 0.3).all() and (r[3:6] < 8).all() and pcr[0] + 1 < r[0] < pcr[3] - 1 and pcr[1] + 1 < r[1] < pcr[4] - 1 and \
                pcr[2] + 0.5 < r[2] < pcr[5] - 0.5
            if ok and all(np.hypot(*(r[:2] - g[:2])) > 3.0 for g in gt[b, :k]):
                gt[b, k, :7] = r + np.concatenate([rng.uniform(-0.06, 0.06, 3), r[3:6] * rng.uniform(-0.04, 0.04, 3), rng.uniform(-0.04, 0.04, 1)])
                gt[b, k, 7] = lab
                k += 1
            if k == 8:
                break
        assert k >= 4, k
    out['pv_gt'] = gt
    model.load_state_dict(pv_seeded_state(model))        # (pass 1 advanced the running statistics)
    batch = make_batch(gt)
    sampled = []
    orig = ProposalTargetLayer.subsample_rois

    def recording(self, max_overlaps):
        idx = orig(self, max_overlaps)
        sampled.append(idx.clone())
        return idx
    ProposalTargetLayer.subsample_rois = recording
    inter = {}
    orig_pool = model.roi_head.roi_grid_pool

    def pool_capture(bd):
        inter['point_features'] = bd['point_features'].detach().clone()
        inter['point_cls_scores'] = bd['point_cls_scores'].detach().clone()
        inter['point_coords'] = bd['point_coords'].detach().clone()
        p = orig_pool(bd)
        inter['pooled'] = p.detach().clone()
        return p
    model.roi_head.roi_grid_pool = pool_capture
    model.roi_head.proposal_layer = capture
    np.random.seed(5)
    torch.manual_seed(5)
    try:
        ret, tb, _ = model(batch)
    finally:
        ProposalTargetLayer.subsample_rois = orig
        model.roi_head.roi_grid_pool, model.roi_head.proposal_layer = orig_pool, orig_pl
    # intermediate results (first-stage proposals before sampling, keypoints, RoI-grid pooled features: a failing comparison
    # of the second-stage outputs can be traced to the module that differs)
    out['pv_proposals'], out['pv_proposal_labels'] = _np(captured['rois']), _np(captured['labels'])
    out['pv_point_coords'] = _np(inter['point_coords'])
    out['pv_point_features'] = _np(inter['point_features'])[:, :32].copy()
    out['pv_point_cls_scores'] = _np(inter['point_cls_scores'])
    out['pv_pooled'] = _np(inter['pooled'])[:, ::27, :16].copy()          # (256 RoIs, 8 of 216 grid points, 16 of 128 channels)
    loss = ret['loss']
    model.zero_grad()
    loss.backward()
    out['pv_loss'] = np.array([float(loss.detach())])
    out['pv_tb_keys'] = np.array(sorted(tb.keys()))
    out['pv_tb_vals'] = np.array([float(tb[k]) for k in sorted(tb.keys())], np.float64)
    out['pv_sampled'] = np.stack([_np(s) for s in sampled]).astype(np.int64)
    out['pv_rcnn_cls'], out['pv_rcnn_reg'] = _np(ret['rcnn_cls']), _np(ret['rcnn_reg'])
    out['pv_rcnn_cls_gt'], out['pv_rcnn_reg_gt'] = _np(ret['rcnn_cls_gt']), _np(ret['rcnn_reg_gt'])
    out['pv_rois'] = _np(model.roi_head.forward_ret_dict['rois'])
    params = dict(model.named_parameters())
    for n, sl in pv_grads(kind).items():
        out['pv_grad/' + n] = _np(params[n].grad)[sl].copy()
        out['pv_gradmax/' + n] = np.array([float(params[n].grad.abs().max())])
    print('  loss %.5f' % float(loss), {k: round(float(v), 5) for k, v in tb.items()})
    print('  voxels', len(coords), 'sampled', out['pv_sampled'].shape, 'fg rois', int((out['pv_rcnn_cls_gt'] > 0.5).sum()))
    assert np.isfinite(out['pv_loss']).all() and all(np.abs(out['pv_grad/' + n]).max() > 0 for n in pv_grads(kind))


if __name__ == '__main__':
    import_reference()
    only = sys.argv[1:] 
    for name, fn in (('ref_utils.npz', gen_utils), ('ref_anchor_head.npz', gen_head), ('ref_bev_vfe.npz', gen_bev), ('ref_roi_head.npz', gen_roi_head),
                     ('ref_strategies.npz', gen_strategies), ('ref_data_processor.npz', gen_data_processor),
                     ('ref_post_processing.npz', gen_post_processing), ('ref_glue.npz', gen_glue),
                     ('ref_badge.npz', gen_badge), ('ref_partA2.npz', gen_partA2),
                     ('ref_point_path.npz', gen_point_path), ('ref_pvrcnn_detector.npz', gen_pvrcnn_detector),
                     ('ref_pvrcnn_detector_waymo.npz', lambda d: gen_pvrcnn_detector(d, 'waymo'))):
        if only and name not in only:
            continue
        d = {}
        fn(d)
        save(name, d)
    if not only or 'ref_points_in_boxes.npz' in only:
        gen_points_in_boxes_ref()
