#!/usr/bin/env python
"""GPU: what torch.use_deterministic_algorithms(True) costs a training step (forward + backward, 16 frames x 20,000 points, no
optimizer): SECOND and PV-RCNN, default mode against deterministic mode, same process."""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    B = 16
    pts, off, gt = kitti_batch(100, B, 20000)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))
    for name, cfg in (('SECOND', second_cfg()), ('PV-RCNN', pv_rcnn_cfg())):
        torch.manual_seed(0)
        model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).train()
        for det in (False, True, False, True):
            torch.use_deterministic_algorithms(det, warn_only=True)
            ts = []
            for rep in range(7):
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
            print('%-8s %-18s %.1f ms per forward + backward (median of the last 5 of 7)' % (name, 'deterministic mode' if det else 'default mode',
                                                                                           1e3 * float(np.median(ts[2:]))), flush=True)
        torch.use_deterministic_algorithms(False)
