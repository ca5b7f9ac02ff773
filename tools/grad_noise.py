#!/usr/bin/env python
"""How much of the sparse-layer gradient difference between the HIP step and the CPU oracle step is rounding noise of the
DENSE half amplified by cancellation? The same SECOND step (same frames, same parameters) is run twice on the GPU with the
BEV part in channels_last and in NCHW storage (other MIOpen kernels, other summation orders, identical mathematics); the
relative L2 distance between the two sets of sparse weight gradients is the floor any comparison against another
implementation of the dense half can reach.  usage: tools/grad_noise.py [kitti|waymo] [B] [points]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np
import torch


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'waymo'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n_points = int(sys.argv[3]) if len(sys.argv) > 3 else (160000 if kind == 'waymo' else 20000)
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    from pcdet.models.backbones_2d.map_to_bev import height_compression as hc
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    ds = SyntheticDataset(num_frames=B, kind=kind, n_points=n_points)
    model = build_network(second_cfg(kind).MODEL, 3, ds).to(dev)
    model.train()
    pts, off, gt = kitti_batch(0, B, n_points, waymo=(kind == 'waymo'))
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))[:, None]
    pick = {'head': lambda m: m.dense_head.conv_cls.weight, 'conv_out': lambda m: m.backbone_3d.conv_out[0].weight,
            'conv_input': lambda m: m.backbone_3d.conv_input[0].weight}
    grads = []
    for cl in (True, False, True):
        hc.CHANNELS_LAST = cl
        model.zero_grad(set_to_none=True)
        batch = {'points': torch.from_numpy(np.concatenate([bidx, pts], 1)).to(dev),
                 'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
                 'batch_size': B}
        ret, tb, _ = model(batch)
        ret['loss'].backward()
        grads.append({k: f(model).grad.double().clone() for k, f in pick.items()})
        print('channels_last=%s loss %.7f' % (cl, float(ret['loss'].detach())))
    for k in pick:
        d01 = float((grads[0][k] - grads[1][k]).norm() / grads[0][k].norm())
        d02 = float((grads[0][k] - grads[2][k]).norm() / grads[0][k].norm())
        print('%-10s channels_last vs NCHW: %.2e   channels_last vs itself: %.2e   |g| %.3e' % (k, d01, d02, float(grads[0][k].norm())))


if __name__ == '__main__':
    main()
