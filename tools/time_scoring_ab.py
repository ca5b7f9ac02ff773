#!/usr/bin/env python
"""GPU, measurement library: the resident CRB stage-1 scoring pass (8 batches of B frames, argv[1], default 16) with the Winograd
product kernel (variant 1: weight fragments in registers) against the LDS-DMA form (variant 3), interleaved, three rounds; third row: the product Winograd kernel with the round-2 farthest-point sampling kernel."""
import os
import sys
import time
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import lib
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    pool = SyntheticDataset(num_frames=10 * B, first_frame=5000, n_points=20000, training=False)
    lab = SyntheticDataset(num_frames=2, n_points=20000)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2), build_synthetic_dataloader(pool, B, workers=12), 0, '/tmp', cfg)
    batches = list(strat.upload_pool_batches(list(range(10 * B)), B))
    strat.score_device_batches(batches[:2])
    torch.cuda.synchronize()
    for rep in range(3):
        for v, fv, name in ((1, 2, 'U in registers'), (3, 2, 'U through LDS-DMA'), (1, 1, 'U in regs, old FPS')):
            lib.crb_winograd4_set_variant(v)
            lib.crb_fps_set_variant(fv)
            strat.score_device_batches(batches[:2])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            strat.score_device_batches(batches[2:])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print('%d frames per batch, %-18s %.1f frames/s (%.1f ms per batch)' % (B, name, 8 * B / dt, dt / 8 * 1e3), flush=True)
    lib.crb_winograd4_set_variant(1)
    lib.crb_fps_set_variant(2)
