#!/usr/bin/env python
"""Torch-native (at::native / rocclr) kernel launches of ONE CRB stage-1 scoring pass (eval-mode PV-RCNN + 5 MC-dropout head passes +
records), attributed to the module scope that launched them (record_function per module; ops outside any module forward are listed
by op name).  usage: python tools/prof_glue_scoring.py [frames per batch = 16] [--top N]"""
import collections
import os
import sys
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 60
    pool = SyntheticDataset(num_frames=4 * B, first_frame=5000, n_points=20000, training=False)
    lab = SyntheticDataset(num_frames=2, n_points=20000)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, B, workers=4), 0, '/tmp', cfg)
    batches = list(strat.upload_pool_batches(list(range(4 * B)), B))
    strat.score_device_batches(batches[:2])
    torch.cuda.synchronize()
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
    # functions outside the modules that launch many small kernels, as scopes of their own (module globals: looked up at call time)
    def scope_fn(mod, fname, tag):
        fn = getattr(mod, fname)

        def wrapped(*a, **k):
            with torch.autograd.profiler.record_function('mod:' + tag):
                return fn(*a, **k)
        setattr(mod, fname, wrapped)
    from pcdet.models.detectors import post_processing as pp
    from pcdet.query_strategies import scoring as sc
    for fname in ('final_nms_batched', 'label_entropy', '_first_hit_counts', '_frame_points', 'gt_point_stats_device'):
        scope_fn(pp, fname, 'records.' + fname)
    scope_fn(sc, 'pack_records', 'records.pack_records')
    rh = getattr(model, 'roi_head', None)
    if rh is not None:
        for meth in ('proposal_layer', 'roi_grid_pool'):
            fn = getattr(rh, meth)
            setattr(rh, meth, (lambda f, t: (lambda *a, **k: (lambda rf: (rf.__enter__(), f(*a, **k), rf.__exit__(None, None, None))[1])(
                torch.autograd.profiler.record_function('mod:' + t))))(fn, 'roi_head.' + meth))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        strat.score_device_batches(batches[2:3])
        torch.cuda.synchronize()
    native = lambda n: ('at::native' in n) or ('rocclr' in n) or n.startswith('Memcpy') or n.startswith('Memset')
    by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    total = [0, 0.0]
    heavy = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        ks = [k for k in getattr(e, 'kernels', []) if native(k.name)]
        if not ks or any(getattr(c, 'kernels', None) for c in e.cpu_children):
            continue
        p = e.cpu_parent
        while p is not None and not p.name.startswith('mod:'):
            p = p.cpu_parent
        site = ('in ' + p.name[4:]) if p is not None else 'outside the modules'
        heavy[(site, e.name.replace('aten::', ''))][0] += len(ks)
        heavy[(site, e.name.replace('aten::', ''))][1] += sum(k.duration for k in ks)
        rec = by_site[site]
        rec[0] += len(ks)
        rec[1] += sum(k.duration for k in ks)
        rec[2][e.name.replace('aten::', '')] += len(ks)
        total[0] += len(ks)
        total[1] += sum(k.duration for k in ks)
    print('scoring pass of %d frames: %d torch-native kernel launches, %.2f ms of device time' % (B, total[0], total[1] / 1e3))
    for site, (n, us, ops) in sorted(by_site.items(), key=lambda kv: -kv[1][0])[:top]:
        print('%4d launches %7.1f us  %-46s [%s]' % (n, us, site[:46], ', '.join('%s x%d' % (k, v) for k, v in ops.most_common(9))))
    print('heaviest (site, op) pairs by device time:')
    for (site, op), (n, us) in sorted(heavy.items(), key=lambda kv: -kv[1][1])[:25]:
        print('  %7.1f us  x%-3d %-14s %s' % (us, n, op, site))
