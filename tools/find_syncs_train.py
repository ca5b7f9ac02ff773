#!/usr/bin/env python
"""List the host synchronisations inside one PV-RCNN (or SECOND) training step (torch's sync debug mode).
usage: python tools/find_syncs_train.py [pvrcnn|second]"""
import os, sys, traceback, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch  # noqa: E402

if __name__ == '__main__':
    import bench
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg, pv_rcnn_cfg
    from pcdet.models import build_network
    which = sys.argv[1] if len(sys.argv) > 1 else 'pvrcnn'
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg() if which == 'pvrcnn' else second_cfg()
    model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2, n_points=args.points)).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    batches = bench.make_batches(args, 0, dev)
    for b in batches:
        b['point_frame_counts_host'] = np.diff(b['point_frame_offsets'].cpu().numpy()).tolist()

    def step(i):
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(dict(batches[i % len(batches)]))
        loss = ret['loss'].mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    seen = []

    def show(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack()[:-1] if 'crb-active-3ddet_amd' in f.filename]
        seen.append(' <- '.join('%s:%d %s' % (os.path.relpath(f.filename, ROOT), f.lineno, f.name) for f in reversed(st[-4:])))
    warnings.showwarning = show
    warnings.simplefilter('always')
    torch.cuda.set_sync_debug_mode('warn')
    step(3)
    torch.cuda.set_sync_debug_mode('default')
    print('%d synchronising calls in one %s training step:' % (len(seen), which))
    for s in seen:
        print('  ', s)
