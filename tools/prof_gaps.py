#!/usr/bin/env python
"""Where the GPU waits for the host in one training step (idle 7.1 ms in 76 gaps > 20 us per PV-RCNN step).
One step under torch.profiler (CPU + device activities, module scopes as in tools/prof_glue.py); every interval with no kernel on
any stream longer than --min-us is attributed to what the HOST was doing when the gap began: the innermost module scope /
autograd node / top-level op whose CPU interval contains the gap's start. Prints the idle time by owner and the longest gaps.
usage: python tools/prof_gaps.py [--model pvrcnn|second] [--min-us 20] [--top 40]"""
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
    ap.add_argument('--min-us', type=float, default=20.0)
    ap.add_argument('--top', type=int, default=40)
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
        with torch.autograd.profiler.record_function('step:zero_grad'):
            opt.zero_grad(set_to_none=True)
        with torch.autograd.profiler.record_function('step:forward'):
            ret, tb, _ = model(dict(batch))
        with torch.autograd.profiler.record_function('step:backward'):
            ret['loss'].backward()
        with torch.autograd.profiler.record_function('step:clip+adamw'):
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    from torch.autograd import DeviceType
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        step()
        torch.cuda.synchronize()
    ev = list(prof.events())
    # device-side events also carry the record_function ranges (gpu_user_annotation): kernels only
    def is_kernel(e):
        n = e.name
        return e.device_type == DeviceType.CUDA and not (n.startswith('mod:') or n.startswith('step:') or 'Backward' in n or
                                                         n.startswith('autograd::') or n.startswith('aten::') or
                                                         n.startswith('Optimizer') or n.startswith('ProfilerStep'))
    kern = sorted([(e.time_range.start, e.time_range.end, e.name) for e in ev if is_kernel(e)], key=lambda t: t[0])
    cpu = [e for e in ev if e.device_type == DeviceType.CPU]
    # the second step only (steady state): from its zero_grad scope to the last kernel
    z = sorted([e for e in cpu if e.name == 'step:zero_grad'], key=lambda e: e.time_range.start)
    t_begin = z[-1].time_range.start
    kern = [k for k in kern if k[0] >= t_begin]
    owners = [e for e in cpu if e.time_range.end >= t_begin and (e.name.startswith('mod:') or e.name.startswith('step:') or
                                                                 'Backward' in e.name or e.name.startswith('autograd::'))]
    owners.sort(key=lambda e: (e.time_range.start, -e.time_range.end))

    def owner_at(t):
        best = None
        for e in owners:
            if e.time_range.start > t:
                break
            if e.time_range.end >= t:
                best = e                      # later start = more deeply nested
        return best.name if best is not None else '(no scope)'
    busy_end = kern[0][1]
    gaps = []
    for s, e, n in kern[1:]:
        if s - busy_end > a.min_us:
            gaps.append((s - busy_end, busy_end, n))
        busy_end = max(busy_end, e)
    if os.environ.get('PROF_GAPS_DEBUG'):
        for s_, e_, n_ in sorted(kern, key=lambda k: k[0] - k[1])[:12]:
            print('   longest device range %.1f us %s' % (e_ - s_, n_[:90]))
    wall = kern[-1][1] - kern[0][0]
    idle = sum(g[0] for g in gaps)
    print('%s step: %.2f ms from first to last kernel, %d kernels, idle %.2f ms in %d gaps > %.0f us' % (
        a.model, wall / 1e3, len(kern), idle / 1e3, len(gaps), a.min_us))
    by = collections.defaultdict(lambda: [0.0, 0])
    for d, t, n in gaps:
        o = owner_at(t + 1.0)
        by[o][0] += d
        by[o][1] += 1
    print('idle time by what the host was in when the gap began:')
    for o, (d, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print('  %8.1f us  %3d gaps  %s' % (d, c, o))
    print('longest gaps:')
    for d, t, n in sorted(gaps, key=lambda g: -g[0])[:15]:
        print('  %8.1f us  in %-60s before %s' % (d, owner_at(t + 1.0)[:60], n[:70]))


if __name__ == '__main__':
    main()
