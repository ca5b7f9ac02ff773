"""Deterministic mode (torch.use_deterministic_algorithms): the weight gradients of the BEV backbone's two strided convolutions come
from the gather-GEMM weight-gradient kernel on a dense pair list (crbhip/dense_strided.py) instead of MIOpen's atomically-summed
split-K kernel; base_bev_backbone.py:33-37,50-55. CPU: the pair lists against autograd. GPU: the gradients against f64, and a SECOND
training step run three times gives bit-equal losses, gradients and running statistics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize('kind,shape', [('conv', (2, 3, 9, 8, 3, 3, 2, 1)), ('conv', (1, 2, 7, 7, 3, 3, 1, 1)), ('conv', (2, 2, 8, 6, 2, 2, 2, 0)),
                                        ('deconv', (2, 3, 5, 4, 2, 2, 2, 0)), ('deconv', (1, 2, 4, 4, 1, 1, 1, 0))])
def test_dense_pair_lists_reproduce_the_weight_gradient(kind, shape):
    from crbhip import dense_strided
    N, C, H, W, kh, kw, s, p = shape
    K = 4
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W, dtype=torch.float64)
    w = torch.randn((K, C, kh, kw) if kind == 'conv' else (C, K, kh, kw), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, s, p) if kind == 'conv' else F.conv_transpose2d(x, w, None, s, 0)
    dy = torch.randn_like(y)
    y.backward(dy)
    pin, pout, start = dense_strided._pairs(kind, N, H, W, kh, kw, s, p, torch.device('cpu'))
    assert start.tolist()[0] == 0 and start.tolist()[-1] == pin.numel() == pout.numel() and len(start) == kh * kw + 1
    xr = x.permute(0, 2, 3, 1).reshape(-1, C)
    dyr = dy.permute(0, 2, 3, 1).reshape(-1, K)
    for k in range(kh * kw):
        a, b = start[k].item(), start[k + 1].item()
        assert bool((pout[a + 1:b] > pout[a:b - 1]).all())                          # ascending output rows inside a tap
        dwk = xr[pin[a:b].long()].t() @ dyr[pout[a:b].long()]                        # (C, K)
        want = w.grad[:, :, k // kw, k % kw].t() if kind == 'conv' else w.grad[:, :, k // kw, k % kw]
        np.testing.assert_allclose(dwk.numpy(), want.numpy(), rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize('kind,cin,cout,H,W,k,s,p', [('conv', 128, 256, 40, 36, 3, 2, 1), ('deconv', 256, 256, 20, 18, 2, 2, 0),
                                                    ('conv', 64, 64, 21, 17, 3, 2, 1), ('deconv', 128, 256, 9, 11, 2, 2, 0)])
def test_deterministic_weight_gradient_against_f64(dev, kind, cin, cout, H, W, k, s, p):
    from crbhip import dense_strided
    torch.manual_seed(1)
    x = torch.randn(2, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cout, cin, k, k) if kind == 'conv' else (cin, cout, k, k), device=dev) / np.sqrt(cin * k * k)).requires_grad_(True)
    conv = (torch.nn.Conv2d(cin, cout, k, s, p, bias=False) if kind == 'conv' else torch.nn.ConvTranspose2d(cin, cout, k, s, bias=False)).to(dev)
    with torch.no_grad():
        conv.weight.copy_(w)
    assert dense_strided.supported(conv, x)
    xg = x.clone().requires_grad_(True)
    y = dense_strided.conv_det(conv, xg)
    dy = torch.randn_like(y)
    y.backward(dy)
    x64, w64 = x.double().requires_grad_(True), w.detach().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p) if kind == 'conv' else F.conv_transpose2d(x64, w64, None, s, 0)
    y64.backward(dy.double())
    rel = lambda got, want: float((got.double() - want).abs().max() / want.abs().max())
    assert rel(y, y64.detach()) <= 1e-5
    assert rel(conv.weight.grad, w64.grad) <= 1e-5, rel(conv.weight.grad, w64.grad)
    assert rel(xg.grad, x64.grad) <= 1e-5
    g0 = conv.weight.grad.clone()
    for _ in range(3):                                     # the same bits every time
        conv.weight.grad = None
        dense_strided.conv_det(conv, x).backward(dy)
        assert torch.equal(conv.weight.grad, g0)


def _three_steps(dev, cfg, deterministic, B=4, reps=3):
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    pts, off, gt = kitti_batch(100, B, 20000)
    bidx = np.repeat(np.arange(B, dtype=np.float32), np.diff(off))
    was = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(deterministic, warn_only=True)
    try:
        runs = []
        for rep in range(reps):
            model.load_state_dict(state)
            ptl = getattr(getattr(model, 'roi_head', None), 'proposal_target_layer', None)
            if ptl is not None:                                    # the RoI sampler's draws: same stream every run
                ptl.generator = torch.Generator(device=dev)
                ptl.generator.manual_seed(1234)
            torch.manual_seed(7)
            b = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev), 'point_frame_offsets': torch.from_numpy(off).to(dev),
                 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': B, 'point_frame_counts_host': np.diff(off).tolist(),
                 'frame_id': np.array(['%06d' % (100 + i) for i in range(B)])}
            ret, tb, _ = model(b)
            model.zero_grad(set_to_none=True)
            ret['loss'].backward()
            torch.cuda.synchronize()
            runs.append((float(ret['loss'].detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                         {k: v.clone() for k, v in model.state_dict().items() if 'running' in k}))
    finally:
        torch.use_deterministic_algorithms(was)
    return runs


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['second', 'pv_rcnn'])
def test_training_step_is_bit_reproducible_in_deterministic_mode(dev, name):
    """loss, every parameter gradient and every running statistic of three runs of one step are bit-equal (SECOND: 84 gradients;
    PV-RCNN: 189 - the set-abstraction scatter adds in 64-bit fixed point, the BEV interpolation's backward and the two strided
    convolutions' weight gradients leave the atomics); and the mode's gradients are the default mode's up to summation order."""
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    cfg = second_cfg() if name == 'second' else pv_rcnn_cfg()
    runs = _three_steps(dev, cfg, True)
    assert len(runs[0][1]) >= (84 if name == 'second' else 189)
    for r in runs[1:]:
        assert r[0] == runs[0][0]
        differ = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], r[1][n])]
        assert not differ, differ
        assert all(torch.equal(runs[0][2][k], r[2][k]) for k in runs[0][2])
    ref = _three_steps(dev, cfg, False, reps=1)[0]
    assert abs(ref[0] - runs[0][0]) <= 1e-5 * abs(ref[0])
    for n, g in runs[0][1].items():
        err = float((g - ref[1][n]).abs().max()) / max(float(ref[1][n].abs().max()), 1e-30)
        assert err <= 2e-4, (n, err)
