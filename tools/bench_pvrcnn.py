#!/usr/bin/env python
"""PV-RCNN fwd+bwd+AdamW on synthetic KITTI frames (BASELINE configs[2]; a parity-test configuration, timed here for
DESIGN.md, not a bench.py line). Usage: python tools/bench_pvrcnn.py [--batch 16] [--steps 5] [--warmup 2]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--points', type=int, default=20000)
    a = ap.parse_args()
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    batches = []
    for k in range(2):
        pts, off, gt = kitti_batch(100 + k * a.batch, a.batch, a.points)
        bidx = np.repeat(np.arange(a.batch, dtype=np.float32), np.diff(off))
        batches.append({'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
                        'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
                        'batch_size': a.batch, 'point_frame_counts_host': np.diff(off).tolist()})

    ahead = {}
    prefetch = os.environ.get('CRB_SPARSE_PREFETCH', '1') == '1'

    # measurement knob: N extra one-element kernels after the forward pass (what does a tiny launch cost a device-bound step?)
    extra = int(os.environ.get('CRB_BENCH_EXTRA_LAUNCHES', '0'))
    tick = torch.zeros((1,), device=dev)

    def step(i):
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(ahead.pop(i, None) or dict(batches[i % 2]))
        for _ in range(extra):
            tick.add_(1.0)
        if prefetch:
            ahead.clear()
            ahead[i + 1] = model.prefetch_sparse(dict(batches[(i + 1) % 2]))
        ret['loss'].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return ret['loss']

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'metric': 'frames/s PV-RCNN fwd+bwd+AdamW, synthetic KITTI 20k-pt clouds', 'value': round(a.batch * a.steps / dt, 2),
                      'ms_per_step': round(1e3 * dt / a.steps, 2), 'batch': a.batch, 'steps': a.steps, 'dtype': 'f32',
                      'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 1), 'loss': round(float(loss), 4)}))


if __name__ == '__main__':
    main()
