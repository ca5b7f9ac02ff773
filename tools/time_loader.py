#!/usr/bin/env python
"""Where the through-loader CRB scoring pass loses time: (1) one frame generated in-process, (2) the pool loader alone
(frames/s the worker processes deliver to the main process), (3) loader + upload, (4) loader + upload + scoring.
usage: python tools/time_loader.py [frames] [workers]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.datasets import synthetic as syn
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    from pcdet.utils.common_utils import effective_cpu_count
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else max(2, min(48, effective_cpu_count() - 2))
    print('host: %d cpus (%d usable), torch threads %d, loader workers %d' % (os.cpu_count(), effective_cpu_count(), torch.get_num_threads(), workers))
    t = time.perf_counter()
    for i in range(8):
        syn.kitti_frame(9000 + i, 20000)
    print('one frame generated in-process: %.1f ms' % ((time.perf_counter() - t) / 8 * 1e3))
    dev = torch.device('cuda', 0)
    torch.set_num_threads(2)
    cfg = pv_rcnn_cfg()
    pool = SyntheticDataset(num_frames=n, first_frame=5000, training=False)
    lab = SyntheticDataset(num_frames=2)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, 16, workers=workers),
                           0, '/tmp', cfg)
    idx = list(range(n))
    strat.score_pool(idx[:32], 16)                                     # fork the workers, MIOpen search
    torch.cuda.synchronize()
    t = time.perf_counter()
    k = 0
    first = None
    for b in strat.iter_pool_batches(idx, 16):
        k += b['batch_size']
        if first is None:
            first = time.perf_counter() - t
    dt = time.perf_counter() - t
    print('loader alone: %.0f frames/s (%d frames in %.2f s, first batch after %.2f s)' % (k / dt, k, dt, first))
    t = time.perf_counter()
    k = 0
    for b in strat.upload_pool_batches(idx, 16):
        k += b['batch_size']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print('loader + upload: %.0f frames/s' % (k / dt))
    kept = []

    def gen():
        for b in strat.upload_pool_batches(idx, 16):
            kept.append(b)
            yield b
    t = time.perf_counter()
    strat.score_device_batches(gen())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print('loader + upload + scoring: %.0f frames/s' % (n / dt))
    t = time.perf_counter()
    strat.score_device_batches(kept)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print('resident scoring: %.0f frames/s' % (n / dt))
    strat.close()


if __name__ == '__main__':
    main()
