"""GPU: every Winograd launch of a REAL SECOND training step checked where it happens: forward, input
gradient and weight gradient of each stride-1 3x3 convolution of the BEV backbone against an f64 convolution of the very tensors
the kernel was handed (crbhip.selfcheck.WinogradInSitu) - no BatchNorm stack between the kernel and the yardstick, so a bug of
relative size 1e-4 in any one layer's dx / dw would show, which the model-level pins (1e-2 through eleven train-mode BatchNorms at
B = 2) cannot promise. Bound 2e-5 of the largest entry (the kernel-level bar of tests/test_winograd_gpu.py); observed 2e-7 .. 2e-6."""
import numpy as np
import pytest
import torch

from synth import kitti_batch

pytestmark = pytest.mark.gpu


def _second_step(dev, B, first=0, n_points=20000):
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(second_cfg('kitti').MODEL, 3, SyntheticDataset(num_frames=B)).to(dev).train()
    pts, off, gt = kitti_batch(first, B, n_points)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))[:, None]
    batch = {'points': torch.from_numpy(np.concatenate([bidx, pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
             'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B}
    return model, batch


@pytest.mark.parametrize('B,every', [(2, 1), (16, 4)])
def test_every_winograd_launch_of_a_second_step_matches_its_f64_convolution(dev, B, every):
    """B = 2: all 11 layers x (forward, dx, dw); B = 16 (the bench batch: 16 x 200 x 176 maps): every fourth launch of each kind
    (the f64 yardsticks of the full-size maps take seconds each)"""
    from crbhip.selfcheck import WinogradInSitu
    from pcdet.models.backbones_2d import base_bev_backbone as bev
    assert bev.WINOGRAD, 'the Winograd path is the default'
    model, batch = _second_step(dev, B)
    with WinogradInSitu(every=every) as chk:
        ret, tb, _ = model(batch)
        ret['loss'].backward()
    torch.cuda.synchronize()
    print('B=%d: %s' % (B, chk.summary()))
    n = {k: sum(1 for r in chk.records if r[0] == k) for k in ('fwd', 'dgrad', 'wgrad')}
    want = 11 if every == 1 else 3
    assert n['fwd'] >= want and n['dgrad'] >= want and n['wgrad'] >= want, n
    chk.assert_all(2e-5)
    # the hooks are gone afterwards
    from crbhip import winograd
    assert winograd.conv3x3.__module__ == 'crbhip.winograd'
