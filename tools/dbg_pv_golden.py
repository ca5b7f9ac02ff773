import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/crb-active-3ddet_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from golden.make_goldens import PV_FIRST_FRAME, PV_GRADS, PV_KEYPOINTS, PV_POINTS, pv_seeded_state
from pcdet.datasets import SyntheticDataset
from pcdet.model_cfgs import pv_rcnn_cfg
from pcdet.models import build_network
from test_pvrcnn_gpu import _batch
dev = torch.device('cuda', 0)
G = np.load('/root/repo/tests/golden/ref_pvrcnn_detector.npz')
cfg = pv_rcnn_cfg().MODEL
cfg.PFE.NUM_KEYPOINTS = PV_KEYPOINTS; cfg.POINT_HEAD.NUM_KEYPOINTS = PV_KEYPOINTS; cfg.ROI_HEAD.DP_RATIO = 0.0
model = build_network(cfg, 3, SyntheticDataset(num_frames=2))
model.load_state_dict(pv_seeded_state(model)); model.to(dev).train()
b, *_ = _batch(dev, PV_FIRST_FRAME, 2, PV_POINTS)
b['gt_boxes'] = torch.from_numpy(G['pv_gt']).to(dev)
# the reference's sampled RoIs as boxes (equal-score proposals have no defined order, a few neighbours come out swapped)
ref_sampled = np.take_along_axis(G['pv_proposals'], G['pv_sampled'][:, :, None], axis=1)
model.roi_head.proposal_target_layer.injected_rois = torch.from_numpy(ref_sampled)
inter = {}
head = model.roi_head
orig_pl, orig_pool = head.proposal_layer, head.roi_grid_pool
def pl(bd, nms_config):
    t = orig_pl(bd, nms_config=nms_config)
    inter['proposals'], inter['labels'] = bd['rois'].detach().clone(), bd['roi_labels'].detach().clone()
    return t
def pool(bd):
    inter['pf'], inter['ps'], inter['pc'] = bd['point_features'].detach().clone(), bd['point_cls_scores'].detach().clone(), bd['point_coords'].detach().clone()
    p = orig_pool(bd)
    inter['pooled'] = p.detach().clone()
    return p
head.proposal_layer, head.roi_grid_pool = pl, pool
ret, tb, _ = model(b)
pr = inter['proposals'].cpu().numpy(); d = np.abs(pr - G['pv_proposals']).max(-1)
print('proposals: rows differing > 1e-3:', (d > 1e-3).sum(), 'of', d.shape, 'first differing', np.argwhere(d > 1e-3)[:6].tolist(), 'labels equal', (inter['labels'].cpu().numpy() == G['pv_proposal_labels']).all())
print('point_coords max diff', np.abs(inter['pc'].cpu().numpy() - G['pv_point_coords']).max())
pf = inter['pf'].cpu().numpy()[:, :32]; print('point_features max diff', np.abs(pf - G['pv_point_features']).max(), 'scale', np.abs(G['pv_point_features']).max())
print('point_cls_scores max diff', np.abs(inter['ps'].cpu().numpy() - G['pv_point_cls_scores']).max())
po = inter['pooled'].cpu().numpy()[:, ::27, :16]; dd = np.abs(po - G['pv_pooled']); print('pooled max diff', dd.max(), 'scale', np.abs(G['pv_pooled']).max(), 'rois differing', (dd.reshape(dd.shape[0], -1).max(-1) > 1e-3).sum())
for k, want in zip(G['pv_tb_keys'], G['pv_tb_vals']):
    print('%-20s got %.6f want %.6f' % (k, float(tb[k]), want))
rois = model.roi_head.forward_ret_dict['rois'].cpu().numpy()
d = np.abs(rois - G['pv_rois'])
print('rois max diff', d.max(), 'rows differing > 1e-3:', (d.max(-1) > 1e-3).sum(), 'of', d.shape[:2])
for k, name in (('rcnn_cls', 'pv_rcnn_cls'), ('rcnn_reg', 'pv_rcnn_reg'), ('rcnn_cls_gt', 'pv_rcnn_cls_gt'), ('rcnn_reg_gt', 'pv_rcnn_reg_gt')):
    got, want = ret[k].detach().float().cpu().numpy().reshape(G[name].shape), G[name]
    print(k, 'max diff', np.abs(got - want).max(), 'scale', np.abs(want).max(), 'rows differing', (np.abs(got-want).reshape(got.shape[0], -1).max(-1) > 1e-3*np.abs(want).max()).sum())
