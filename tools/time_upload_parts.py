#!/usr/bin/env python
"""per-batch host time of the through-loader upload path, part by part (loader wait+merge / pinned copy / H2D issue)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np
import torch

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    n = 1600
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    dev = torch.device('cuda', 0)
    torch.set_num_threads(2)
    cfg = pv_rcnn_cfg()
    pool = SyntheticDataset(num_frames=n, first_frame=5000, training=False)
    lab = SyntheticDataset(num_frames=2)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, 16, workers=workers), 0, '/tmp', cfg)
    idx = list(range(n))
    list(strat.iter_pool_batches(idx[:64], 16))
    side = torch.cuda.Stream()
    t_wait = t_pin = t_stage = 0.0
    t0 = time.perf_counter()
    it = strat.iter_pool_batches(idx, 16)
    k = 0
    while True:
        a = time.perf_counter()
        try:
            b = next(it)
        except StopIteration:
            break
        c = time.perf_counter()
        p = strat._pin_batch(b)
        d = time.perf_counter()
        s = strat._stage_batch(p, dev, side)
        strat._pin_ring['events'][p['_pin_slot']] = s[1]
        e = time.perf_counter()
        t_wait += c - a; t_pin += d - c; t_stage += e - d
        k += 1
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print('workers %d: %d batches in %.2f s (%.0f frames/s): per batch loader wait+merge %.1f ms, pinned copy %.1f ms, H2D issue %.1f ms' % (
        workers, k, tot, n / tot, 1e3 * t_wait / k, 1e3 * t_pin / k, 1e3 * t_stage / k))
    strat.close()
