#!/usr/bin/env python
"""GPU: which parameter gradients of a PV-RCNN / SECOND training step differ between two runs of the same step (same weights, same
frames, same RoI-sampler seed)? Prints, per parameter, whether the two gradients are bit-equal and the largest difference relative to
the gradient's largest entry; then the same for the loss and the running statistics. Usage: python tools/dbg_determinism.py [B]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

if __name__ == '__main__':
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    from pcdet.models import build_network
    from pcdet.datasets.synthetic import kitti_batch
    dev = torch.device('cuda', 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    if '--deterministic' in sys.argv:
        torch.use_deterministic_algorithms(True, warn_only=True)
    for name, cfg in (('SECOND', second_cfg()), ('PV-RCNN', pv_rcnn_cfg())):
        torch.manual_seed(0)
        model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).train()
        state = {k: v.clone() for k, v in model.state_dict().items()}
        pts, off, gt = kitti_batch(100, B, 20000)
        bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))
        runs = []
        for rep in range(3):
            model.load_state_dict(state)
            gen = getattr(getattr(getattr(model, 'roi_head', None), 'proposal_target_layer', None), 'generator', None)
            if gen is not None:
                gen.manual_seed(1234)
            torch.manual_seed(7)
            b = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
                 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B, 'point_frame_counts_host': np.diff(off).tolist(),
                 'frame_id': np.array(['%06d' % (100 + i) for i in range(B)])}
            ret, tb, _ = model(b)
            model.zero_grad(set_to_none=True)
            ret['loss'].backward()
            torch.cuda.synchronize()
            runs.append(({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(ret['loss']),
                         {k: v.clone() for k, v in model.state_dict().items() if 'running' in k}))
        print('== %s, %d frames: loss %r %r %r' % (name, B, runs[0][1], runs[1][1], runs[2][1]))
        ndiff = 0
        for n in runs[0][0]:
            a = runs[0][0][n]
            worst = max(float((a - r[0][n]).abs().max()) for r in runs[1:])
            if worst != 0.0:
                ndiff += 1
                print('   %-60s differs: %.2e of its largest entry' % (n, worst / max(float(a.abs().max()), 1e-30)))
        nstat = sum(1 for k in runs[0][2] if any(not torch.equal(runs[0][2][k], r[2][k]) for r in runs[1:]))
        print('   %d of %d parameter gradients differ between runs; %d of %d running statistics differ' % (ndiff, len(runs[0][0]), nstat, len(runs[0][2])))
