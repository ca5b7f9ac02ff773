#!/usr/bin/env python
"""Which host op launches the small torch kernels of a SECOND training step (fills, copies, integer adds ...): one step
under torch.profiler with Python stacks, kernels grouped by (kernel name, innermost repo frame)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch

if __name__ == '__main__':
    import bench
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg, pv_rcnn_cfg
    from pcdet.models import build_network
    which = sys.argv[1] if len(sys.argv) > 1 else 'second'
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    if which == 'score':
        from pcdet.datasets import build_synthetic_dataloader
        from pcdet.query_strategies import build_strategy
        cfg = pv_rcnn_cfg()
        pool = SyntheticDataset(num_frames=48, first_frame=5000, n_points=20000, training=False)
        lab = SyntheticDataset(num_frames=2, n_points=20000)
        model = build_network(cfg.MODEL, 3, pool).to(dev)
        strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, 16), 0, '/tmp', cfg)
        sb = list(strat.upload_pool_batches(list(range(48)), 16))
        strat.score_device_batches(sb[:2])
        torch.cuda.synchronize()
        step = lambda i: strat.score_device_batches(sb[2:3])
        # module-level attribution: every top-level module of the chain and the record packing run under a named range
        from torch.profiler import record_function
        from pcdet.query_strategies import crb_sampling as _cs
        for name, mod in model.named_children():
            mod.register_forward_pre_hook(lambda m, a, _n=name: m.__dict__.__setitem__('_rf', record_function('MOD:' + _n).__enter__()))
            mod.register_forward_hook(lambda m, a, o: m.__dict__.pop('_rf').__exit__(None, None, None))
        _orig = _cs.crb_frame_records
        def _wrapped(*a, **k):
            with record_function('MOD:crb_frame_records'):
                return _orig(*a, **k)
        _cs.crb_frame_records = _wrapped
    cfg = pv_rcnn_cfg() if which == 'pvrcnn' else second_cfg()
    model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2, n_points=args.points)).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    batches = bench.make_batches(args, 0, dev)
    for b in batches:
        b['point_frame_counts_host'] = np.diff(b['point_frame_offsets'].cpu().numpy()).tolist()

    def train_step(i):
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(dict(batches[i % len(batches)]))
        loss = ret['loss'].mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
    if which != 'score':
        step = train_step
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(3)
        torch.cuda.synchronize()
    ev = prof.events()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in ev:
        if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::'):
            continue
        dt = e.self_device_time_total
        lo, hi = (float(os.environ.get('PROF_MIN_US', '0')), float(os.environ.get('PROF_MAX_US', '20')))
        if dt <= lo or dt > hi:
            continue
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 5:
            if q.name.startswith('MOD:'):
                chain = [q.name]
                break
            chain.append(q.name[:48])
            q = q.cpu_parent
        if which == 'score' and chain and not chain[0].startswith('MOD:'):
            q2 = e.cpu_parent
            while q2 is not None:
                if q2.name.startswith('MOD:'):
                    chain = [q2.name]
                    break
                q2 = q2.cpu_parent
        where = ' <- '.join(chain) if chain else '(top level)'
        r = rows[(e.name, where)]
        r[0] += 1
        r[1] += dt
    out = sorted(rows.items(), key=lambda kv: -kv[1][1])
    print('aten ops with own device time in (PROF_MIN_US, PROF_MAX_US] = (0, 20] by default: %d, %.3f ms' % (sum(v[0] for _, v in out), sum(v[1] for _, v in out) / 1e3))
    for (op, where), (n, us) in out[:110]:
        print('%4d x %6.1f us  %-28s %s' % (n, us / n, op, where))
