#!/usr/bin/env python
"""The resident CRB stage-1 scoring pass alone, for rocprofv3 kernel traces: 8 batches of B frames (argv[1], default 16) scored
back to back after a warm-up.  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o sc -- python tools/prof_scoring_resident.py ;
PROF_MARKER=vox_insert PROF_GAPS=30 python tools/prof_summary.py /tmp/p/*/sc_kernel_trace.csv 4"""
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
    if os.environ.get('CRB_FPS_VARIANT'):                 # measurement library (CRB_MEASURE_LIB=1): A/B of the sampling kernel
        from crbhip import lib
        lib.crb_fps_set_variant(int(os.environ['CRB_FPS_VARIANT']))
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
    strat.score_device_batches(batches[2:])
    torch.cuda.synchronize()
