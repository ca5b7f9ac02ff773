#!/usr/bin/env python
"""GPU, measurement library: a PV-RCNN training step (forward + backward, 16 frames x 20,000 points) with the two farthest-point
sampling kernels (2 = fps2_kernel, the product kernel; 1 = the round-2 kernel): does the sampler on the side stream slow the kernels
it shares CUs with?"""
import os
import sys
import time
import numpy as np
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import lib
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    B = 16
    pts, off, gt = kitti_batch(100, B, 20000)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).train()
    for rep in range(3):
        for v in (2, 1):
            lib.crb_fps_set_variant(v)
            ts = []
            for it in range(8):
                b = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
                     'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B, 'point_frame_counts_host': np.diff(off).tolist(),
                     'frame_id': np.array(['%06d' % (100 + i) for i in range(B)])}
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ret, tb, _ = model(b)
                model.zero_grad(set_to_none=True)
                ret['loss'].backward()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print('PV-RCNN forward + backward, sampling kernel %d: %.2f ms (median of the last 6 of 8)' % (v, 1e3 * float(np.median(ts[2:]))), flush=True)
    lib.crb_fps_set_variant(2)
