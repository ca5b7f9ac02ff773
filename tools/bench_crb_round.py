#!/usr/bin/env python
"""One whole CRB selection round at the reference's KITTI budget (SURVEY §8d metric 2): stage 1 over a synthetic pool,
stage 2 on K1*N = 500 frames (bs=1 training-mode passes + kmeans++ to K2*N = 300), stage 3 greedy density balance to
N = 100 frames. Prints seconds per stage. Usage: python tools/bench_crb_round.py [--pool 640] [--select 100]"""
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
    ap.add_argument('--pool', type=int, default=640)
    ap.add_argument('--select', type=int, default=100)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--clustering', default='kmeans++', choices=['kmeans++', 'kmeans++_device'])
    ap.add_argument('--workers', type=int, default=16, help='DataLoader workers of the unlabelled pool loader')
    ap.add_argument('--stage2-batch', type=int, default=16, help='frames per stage-2 pass (1 = the reference loop of bs=1 passes)')
    a = ap.parse_args()
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    dev = torch.device('cuda', 0)
    cfg = pv_rcnn_cfg()
    cfg.ACTIVE_TRAIN.SELECT_NUMS = a.select
    cfg.ACTIVE_TRAIN.ACTIVE_CONFIG.CLUSTERING = a.clustering
    cfg.ACTIVE_TRAIN.ACTIVE_CONFIG.STAGE2_BATCH = a.stage2_batch
    torch.manual_seed(0)
    pool = SyntheticDataset(num_frames=a.pool, first_frame=2000)
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    strat = build_strategy('crb', model, build_synthetic_dataloader(SyntheticDataset(num_frames=2), 2),
                           build_synthetic_dataloader(pool, a.batch, workers=a.workers), 0, '/tmp', cfg)
    strat.score_pool(list(range(2 * a.batch)), a.batch)                       # warm-up: MIOpen solver search, caches
    rec = strat.score_pool(list(range(4)), a.batch)
    strat.grad_embeddings(list(range(4)), rec)
    if a.stage2_batch > 1:
        rec = strat.score_pool(list(range(a.stage2_batch)), a.batch)
        strat.grad_embeddings_batched(list(range(a.stage2_batch)), rec, a.stage2_batch)
    torch.cuda.synchronize()
    strat.profile_stage2 = bool(os.environ.get('CRB_STAGE2_TIMING'))
    t0 = time.perf_counter()
    picked = strat.query()
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    t = strat.timings
    out = {'pool_frames': a.pool, 'K1N': min(strat.k1 * a.select, a.pool), 'K2N': min(strat.k2 * a.select, a.pool),
           'selected': len(picked), 'loader_workers': a.workers, 'stage1_s': round(t['stage1_s'], 3),
           'stage1_frames_per_s': round(a.pool / t['stage1_s'], 1),
           'stage2_s': round(t['stage2_s'], 3), 'stage2_grad_embeddings_s': round(t['stage2_embed_s'], 3),
           'clustering': a.clustering, 'stage2_batch': a.stage2_batch, 'stage2_kmeanspp_s': round(t['stage2_s'] - t['stage2_embed_s'], 3),
           'stage3_s': round(t['stage3_s'], 4), 'round_s': round(total, 3),
           'stage2_wait_for_frames_s': round(t.get('stage2_wait_for_frames_s', -1), 3),
           'stage2_passes_s': round(t.get('stage2_passes_s', -1), 3),
           'data': 'synthetic KITTI-shaped frames, generated on the host by the pool loader\'s workers inside the stage times'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
