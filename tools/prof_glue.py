#!/usr/bin/env python
"""VERDICT r03 item 2 (glue of the PV-RCNN step): which source lines launch the torch-native (at::native / rocclr copy + fill)
kernels of one training step. torch.profiler with Python stacks; every such kernel is attributed to the innermost frame of
this repository on its launching op's stack (forward), or to the autograd node that launched it (backward).
usage: python tools/prof_glue.py [--model pvrcnn|second] [--top 45]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='pvrcnn')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--top', type=int, default=45)
    a = ap.parse_args()
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg() if a.model == 'pvrcnn' else second_cfg()
    model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    pts, off, gt = kitti_batch(100, a.batch, 20000)
    bidx = np.repeat(np.arange(a.batch, dtype=np.float32), np.diff(off))
    batch = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
             'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
             'batch_size': a.batch, 'point_frame_counts_host': np.diff(off).tolist()}

    def step():
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(dict(batch))
        ret['loss'].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    # module scopes (the profiler delivers no Python stacks on this stack): every module's forward runs inside a record_function
    scopes = {}

    def pre(name):
        def f(mod, inp):
            rf = torch.autograd.profiler.record_function('mod:' + name)
            rf.__enter__()
            scopes.setdefault(id(mod), []).append(rf)
        return f

    def post(mod, inp, out):
        scopes[id(mod)].pop().__exit__(None, None, None)
    for name, m in model.named_modules():
        if name:
            m.register_forward_pre_hook(pre(name))
            m.register_forward_hook(post)
    # finer scopes than the modules where one module launches hundreds of small kernels: methods of the RoI head / its target layer
    def scope_method(obj, meth, tag):
        fn = getattr(obj, meth)

        def wrapped(*args, **kw):
            with torch.autograd.profiler.record_function('mod:' + tag):
                return fn(*args, **kw)
        setattr(obj, meth, wrapped)
    rh = getattr(model, 'roi_head', None)
    if rh is not None:
        for meth in ('proposal_layer', 'assign_targets', 'roi_grid_pool', 'get_global_grid_points_of_roi', 'get_loss'):
            if hasattr(rh, meth):
                scope_method(rh, meth, 'roi_head.' + meth)
        # (the target layer's sampling methods are not wrapped: an instance with replaced sampling methods keeps the torch layer,
        #  proposal_target_layer.forward_fused - the listing would show launches the product path does not make)
        scope_method(rh.proposal_target_layer, 'forward', 'roi_head.target_layer.forward')
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    ev = list(prof.events())
    native = lambda n: ('at::native' in n) or ('rocclr' in n) or n.startswith('Memcpy') or n.startswith('Memset')
    by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    total = [0, 0.0]
    # innermost op that owns the kernel: ops with kernels and no child that has kernels
    for e in ev:
        ks = [k for k in getattr(e, 'kernels', []) if native(k.name)]
        if not ks:
            continue
        if any(getattr(c, 'kernels', None) for c in e.cpu_children):
            continue
        site = None
        for fr in (e.stack or []):
            if ('crb-active-3ddet_amd/' in fr) and ('torch/' not in fr):
                site = fr.split('crb-active-3ddet_amd/')[-1]
                break
        if site is None:
            p = e.cpu_parent
            while p is not None and not p.name.startswith('mod:'):
                p = p.cpu_parent
            if p is not None:
                site = 'forward of ' + p.name[4:] + ' | ' + e.name.replace('aten::', '')
        if site is None:
            p = e
            while p is not None and not p.name.startswith('autograd::engine::evaluate_function'):
                p = p.cpu_parent
            site = p.name.replace('autograd::engine::evaluate_function: ', 'backward of ') if p is not None else (
                'optimizer / clip' if any('optim' in fr or 'clip_grad' in fr for fr in (e.stack or [])) else 'other: ' + e.name)
        rec = by_site[site]
        rec[0] += len(ks)
        rec[1] += sum(k.duration for k in ks)
        rec[2][e.name] += len(ks)
        total[0] += len(ks)
        total[1] += sum(k.duration for k in ks)
    print('%s step, bs=%d: %d torch-native kernel launches, %.2f ms of device time' % (a.model, a.batch, total[0], total[1] / 1e3))
    for site, (n, us, ops) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print('%4d launches %7.1f us  %s   [%s]' % (n, us, site[:120], ', '.join('%s x%d' % (k.replace('aten::', ''), v) for k, v in ops.most_common(5))))


if __name__ == '__main__':
    main()
