#!/usr/bin/env python
"""Per-module wall time (with a device sync after each) of the CRB stage-1 scoring forward on one 16-frame batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch
from pcdet.datasets import SyntheticDataset
from pcdet.datasets.synthetic import kitti_batch
from pcdet.model_cfgs import pv_rcnn_cfg
from pcdet.models import build_network
from pcdet.models.detectors.post_processing import crb_frame_records
from pcdet.query_strategies import scoring
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).eval()
for m in model.modules():
    if m.__class__.__name__.startswith('Dropout'):
        m.train()
pts, off, gt = kitti_batch(5000, 16, 20000)
bidx = np.repeat(np.arange(16, dtype=np.float32), np.diff(off))
base = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
        'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': 16, 'point_frame_counts_host': np.diff(off).tolist()}
acc = {}
with torch.no_grad():
    for it in range(6):
        b = dict(base)
        torch.cuda.synchronize(); t_all = time.perf_counter()
        for mod in model.scheduled_modules():
            t0 = time.perf_counter()
            b = mod(b)
            torch.cuda.synchronize()
            if it >= 2:
                acc[type(mod).__name__] = acc.get(type(mod).__name__, 0.0) + time.perf_counter() - t0
        t0 = time.perf_counter()
        r = scoring.pack_records(crb_frame_records(model, b))
        torch.cuda.synchronize()
        if it >= 2:
            acc['crb_frame_records'] = acc.get('crb_frame_records', 0.0) + time.perf_counter() - t0
            acc['TOTAL(serialised)'] = acc.get('TOTAL(serialised)', 0.0) + time.perf_counter() - t_all
for k, v in acc.items():
    print('%-28s %7.2f ms' % (k, 1e3 * v / 4))
