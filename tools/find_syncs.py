#!/usr/bin/env python
"""List the host synchronisations inside one CRB stage-1 scoring pass of a 16-frame batch (torch's sync debug mode): every
`.item()` / `.tolist()` / `.cpu()` drains the launch queue and the GPU then idles until the host has refilled it.
usage: python tools/find_syncs.py [score|train]"""
import os
import sys
import traceback
import warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg()
    pool = SyntheticDataset(num_frames=48, first_frame=5000, n_points=20000, training=False)
    lab = SyntheticDataset(num_frames=2, n_points=20000)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, 16), 0, '/tmp', cfg)
    batches = list(strat.upload_pool_batches(list(range(48)), 16))
    strat.score_device_batches(batches[:2])
    torch.cuda.synchronize()
    seen = []

    def show(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack()[:-1] if 'crb-active-3ddet_amd' in f.filename or 'bench' in f.filename]
        seen.append(' <- '.join('%s:%d %s' % (os.path.relpath(f.filename, ROOT), f.lineno, f.name) for f in reversed(st[-4:])))
    warnings.showwarning = show
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode('warn')
    strat.score_device_batches(batches[2:3])
    torch.cuda.set_sync_debug_mode('default')
    print('%d synchronising calls in one 16-frame scoring pass:' % len(seen))
    for s in seen:
        print('  ', s)
