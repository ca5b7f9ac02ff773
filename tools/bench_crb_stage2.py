#!/usr/bin/env python
"""CRB stage 2 (per-frame gradient embeddings, bs=1 training-mode passes): seconds per frame with the pruned backward and
with the reference's full loss.backward(). Usage: python tools/bench_crb_stage2.py [--frames 24]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=24)
    a = ap.parse_args()
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    dev = torch.device('cuda', 0)
    cfg = pv_rcnn_cfg()
    torch.manual_seed(0)
    pool = SyntheticDataset(num_frames=a.frames, first_frame=900)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(SyntheticDataset(num_frames=2), 2),
                           build_synthetic_dataloader(pool, 8), 0, '/tmp', cfg)
    idx = list(range(a.frames))
    records = strat.score_pool(idx, 8)
    out = {}
    for name, pruned in (('pruned_backward', True), ('full_backward', False)):
        strat.PRUNED_BACKWARD = pruned
        strat.grad_embeddings(idx[:4], records[:4])                  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        strat.grad_embeddings(idx, records)
        torch.cuda.synchronize()
        out[name + '_ms_per_frame'] = round(1e3 * (time.perf_counter() - t0) / a.frames, 2)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
