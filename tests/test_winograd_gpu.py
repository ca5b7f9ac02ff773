"""GPU: hand-written F(2x2,3x3) Winograd f32 convolution (crb_conv3x3_winograd_nhwc, opt-in replacement of MIOpen's implicit
GEMM for the stride-1 3x3 layers of BaseBEVBackbone, base_bev_backbone.py:24-41) against torch's convolution.
Tolerance: 1e-5 of the output scale on unit-scale data; observed 4e-7 against an f64 convolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,C,K,H,W', [(2, 128, 128, 50, 44), (1, 256, 256, 25, 22), (3, 32, 64, 7, 9), (3, 64, 64, 7, 9), (1, 64, 192, 1, 1),
                                        (2, 128, 64, 33, 17)])
def test_winograd_conv_matches_direct_convolution(dev, N, C, K, H, W):
    """forward (+ bias, + ReLU epilogue) and the input gradient as the same kernel on dy; odd sizes exercise the partial
    tiles, 1x1 maps the all-padding patches, (64,192) three channel blocks"""
    from crbhip import winograd
    torch.manual_seed(N * 1000 + C + K + H)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C))
    b = torch.randn(K, device=dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale = float(ref.abs().max())
    U = winograd.weights_forward(w)
    y = winograd.conv3x3_U(x, U, b)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale
    assert torch.equal(winograd.conv3x3_U(x, U, b, relu=True), torch.relu(y))
    assert torch.equal(winograd.conv3x3_U(x, U, b), y)                                   # bitwise reproducible
    # autograd: dx on the Winograd kernel, dw on MIOpen
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True)
    dy = torch.randn_like(y)
    winograd.conv3x3(xg, wg, bg).backward(dy)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.conv2d(x64, w64, b64, padding=1).backward(dy.double())
    for got, want in ((xg.grad, x64.grad), (wg.grad, w64.grad), (bg.grad, b64.grad)):
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_bev_backbone_with_winograd_matches_miopen_path(dev, monkeypatch):
    """BaseBEVBackbone with the opt-in flag: eval (BatchNorm folded, ReLU in the epilogue: one launch per layer) and the
    training forward against an f64 run of the same network, held to the error the default MIOpen path has there. (The
    gradients of the layer are covered by the kernel-level test above, against f64. Whole-backbone gradients are NOT compared
    here: at this small shape the DEFAULT path's gradients differ by up to 2e-2 between runs of the same process —
    tools/dbg_wino.py, with or without the flag — while every hand-written op is bitwise reproducible in isolation and
    MIOpen's convolutions alone move by 1e-7 (tools/dbg_det.py); the full-size SECOND step is checked against the oracle by
    smoke().)"""
    import copy
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import base_bev_backbone as bb
    torch.manual_seed(3)
    cfg = EasyDict({'LAYER_NUMS': [2, 2], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [256, 256]})
    net = bb.BaseBEVBackbone(cfg, 256).to(dev)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    x = torch.randn(2, 256, 48, 40, device=dev).contiguous(memory_format=torch.channels_last)
    outs = {}
    for flag in (False, True):
        monkeypatch.setattr(bb, 'WINOGRAD', flag)
        n2 = copy.deepcopy(net)
        n2.eval()
        with torch.no_grad():
            ev = n2({'spatial_features': x})['spatial_features_2d'].clone()
        n2.train()
        with torch.no_grad():
            tr = n2({'spatial_features': x})['spatial_features_2d'].clone()
        outs[flag] = (ev, tr)
    monkeypatch.setattr(bb, 'WINOGRAD', False)
    n64 = copy.deepcopy(net).double()
    n64.eval()
    with torch.no_grad():
        ev64 = n64({'spatial_features': x.double()})['spatial_features_2d']
        n64.train()
        tr64 = n64({'spatial_features': x.double()})['spatial_features_2d']
    for n, r, a, b in zip(('eval output', 'train output'), (ev64, tr64), outs[False], outs[True]):
        e_m = float((a.double() - r).abs().max() / r.abs().max())
        e_w = float((b.double() - r).abs().max() / r.abs().max())
        print('%-14s error vs f64: MIOpen path %.2e, Winograd path %.2e' % (n, e_m, e_w))
        assert e_w <= max(3.0 * e_m, 2e-5), (n, e_m, e_w)
