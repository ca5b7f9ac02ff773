"""GPU: hand-written F(2x2,3x3) Winograd f32 convolutions (crb_conv3x3_winograd2_nhwc: the default for the stride-1 3x3 layers of
BaseBEVBackbone, base_bev_backbone.py:24-41, instead of MIOpen's implicit GEMM; crb_conv3x3_winograd_nhwc: the round-3 kernel)
against f64 convolutions. Tolerance: 1e-5 of the output scale on unit-scale data forward, 2e-5 input gradient (VERDICT r03 item 1
asks 1e-5 / 1e-4); observed 4e-7 / 6e-7."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('N,C,K,H,W', [(2, 128, 128, 50, 44), (1, 256, 256, 25, 22), (3, 32, 64, 7, 9), (3, 64, 64, 7, 9), (1, 64, 192, 1, 1),
                                        (2, 128, 64, 33, 17)])
def test_first_winograd_kernel_matches_direct_convolution(dev, N, C, K, H, W):
    """round-3 kernel (kept for A/B): forward (+ bias, + ReLU epilogue) and the input gradient as the same kernel on dy; odd
    sizes exercise the partial tiles, 1x1 maps the all-padding patches, (64,192) three channel blocks"""
    from conftest import require_measure_lib
    require_measure_lib()
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
    if winograd.supported(K, C):
        dy = torch.randn_like(y)
        dx = winograd.conv3x3_U(dy, winograd.weights_input_grad(w))
        want = F.conv_transpose2d(dy.double(), w.double(), padding=1)
        assert float((dx.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


# the default kernel (csrc/winograd_conv2.hip). Shapes: the two BEV shapes scaled down; blocks of 16 tile rows that straddle
# images (N * ceil(H/2) not a multiple of 16, H = 5: five image boundaries inside one block); partial tile columns (W = 17, 9,
# 31: ceil(W/2) not a multiple of 4); odd H / W (half tiles); Cin = 8 (ONE chunk: the peeled tail alone), 24 (three), Cout =
# 192 (three channel blocks); more units than CUs (persistent ranges of several units) and fewer
W2_SHAPES = [(2, 128, 128, 50, 44), (1, 256, 256, 25, 22), (3, 64, 64, 7, 9), (20, 24, 64, 5, 9), (2, 8, 192, 33, 17),
             (1, 256, 128, 40, 31), (16, 16, 64, 50, 44)]
# the split-bf16 kernel (csrc/winograd_conv4.hip; the default where it has an instance: Cin % 16 == 0, Cout % 64 == 0, H >= 31).
# Shapes: the BEV shapes scaled down; blocks of 16 tile rows that straddle two images (N * ceil(H/2) not a multiple of 16: the raw
# block's 2-row gap), the smallest map (H = 31: 16 tile rows = one image per block), partial tile columns and odd sizes (W = 9, 5,
# 17, 31: ceil(W/2) not a multiple of 4; H = 33, 37, 63: half tiles), Cin = 16 (ONE chunk) and 48 (three), Cout = 192 (three channel
# blocks), more units than CUs and fewer; (3, 64, 64, 7, 9) and (1, 256, 256, 25, 22): maps below 31 rows - the same calls fall back
# to the f32-MFMA kernel
W4_SHAPES = [(2, 128, 128, 50, 44), (1, 256, 256, 33, 22), (3, 64, 64, 31, 9), (2, 16, 192, 33, 17), (1, 256, 128, 40, 31),
             (16, 16, 64, 50, 44), (5, 32, 64, 37, 5), (3, 48, 128, 63, 70), (3, 64, 64, 7, 9), (1, 256, 256, 25, 22)]


@pytest.mark.parametrize('kernel,N,C,K,H,W', [('f32',) + s for s in W2_SHAPES] + [('x6',) + s for s in W4_SHAPES])
def test_winograd_conv_matches_direct_convolution(dev, monkeypatch, kernel, N, C, K, H, W):
    """forward (+ bias, + ReLU epilogue), the input gradient as the same kernel on dy with the flipped / transposed weight
    image, and the autograd node (dw, db on MIOpen) against an f64 convolution: <= 1e-5 of the output scale forward, <= 2e-5
    of the largest gradient entry; run-to-run bit equality. Both kernels to the same bars: 'f32' = exact-f32 MFMA
    (winograd_conv2.hip), 'x6' = six bf16 MFMA passes over the exact three-way split of the operands (winograd_conv4.hip)"""
    from crbhip import winograd
    monkeypatch.setattr(winograd, 'KERNEL', kernel)
    torch.manual_seed(N * 1000 + C + K + H)
    assert winograd.supported2(C, K, H, W)
    if kernel == 'x6':
        assert winograd._use4(C, K) and hasattr(winograd.weights_forward2(torch.zeros(K, C, 3, 3, device=dev)), 'wino4_shape')
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C))
    b = torch.randn(K, device=dev)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale = float(ref.abs().max())
    imgs = []
    for wt in (w, w.contiguous(memory_format=torch.channels_last)):                     # both weight layouts: same image
        U = winograd.weights_forward2(wt)
        imgs.append(U)
        if kernel == 'f32':
            assert torch.equal(U, winograd.transform_weights2(w.permute(2, 3, 1, 0).contiguous()))
    assert torch.equal(imgs[0], imgs[1])
    y = winograd.conv3x3_U2(x, U, b)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale
    assert torch.equal(winograd.conv3x3_U2(x, U, b, relu=True), torch.relu(y))
    assert torch.equal(winograd.conv3x3_U2(x, U, b), y)                                  # bitwise reproducible
    y0 = winograd.conv3x3_U2(x, U)                                                       # no bias
    assert float((y0.double() - (ref - b.double().view(1, -1, 1, 1))).abs().max()) <= 1e-5 * scale
    dy = torch.randn_like(y)
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.conv2d(x64, w64, b64, padding=1).backward(dy.double())
    if winograd.supported2(K, C, H, W):
        if kernel == 'f32':
            assert torch.equal(winograd.weights_input_grad2(w), winograd.transform_weights2(w.flip(2, 3).permute(2, 3, 0, 1).contiguous()))
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        bg = b.clone().requires_grad_(True)
        winograd.conv3x3(xg, wg, bg).backward(dy)
        for got, want in ((xg.grad, x64.grad), (wg.grad, w64.grad), (bg.grad, b64.grad)):
            assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize('N,C,K,H,W', [(2, 64, 64, 9, 11), (1, 128, 64, 40, 31), (3, 64, 192, 7, 5), (2, 128, 128, 37, 29), (16, 64, 64, 50, 44),
                                        (1, 64, 64, 1, 1), (5, 256, 128, 12, 9)])
def test_winograd_weight_gradient_matches_f64(dev, N, C, K, H, W):
    """crb_winograd2_wgrad (dU = sum over tiles of V (x) M as MFMA GEMMs, dW = G^T dU G) against the f64 weight gradient of the
    direct convolution: <= 2e-5 of the largest entry (observed 1.3e-7; MIOpen's wrw: 1.4e-7 .. 4e-7). Odd sizes: half tiles and
    partial chunks (2 x 4 tiles); 1 x 1 maps: all-padding patches; 16 x 64 x 50 x 44: more chunks than ranges, (1, ...): fewer
    (empty ranges write zero partials). Bit-equal reruns; the gradient lands in the weight's own memory layout."""
    from crbhip import winograd
    torch.manual_seed(N * 100 + C + K + H)
    assert winograd.wgrad_supported(C, K, H, W)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev)
    w64 = w.double().requires_grad_(True)
    F.conv2d(x.double(), w64, None, padding=1).backward(dy.double())
    want = w64.grad
    for like in (w, w.contiguous(memory_format=torch.channels_last)):
        got = winograd.conv3x3_wgrad(x, dy, like)
        assert got.shape == w.shape and got.stride() == like.stride()
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())
        assert torch.equal(got, winograd.conv3x3_wgrad(x, dy, like))
    # the autograd node uses it
    if winograd.supported2(C, K, H, W):
        wg = w.clone().requires_grad_(True)
        winograd.conv3x3(x, wg, None).backward(dy)
        assert torch.equal(wg.grad, winograd.conv3x3_wgrad(x, dy, w))


def test_winograd_kernel_reports_what_it_cannot_run(dev):
    from crbhip import winograd
    assert not winograd.supported2(128, 128, 4, 40)          # H < 5: a block of 16 tile rows would cross > 5 image boundaries
    assert not winograd.supported2(12, 64, 40, 40) and not winograd.supported2(64, 96, 40, 40)
    x = torch.randn(1, 128, 4, 40, device=dev).contiguous(memory_format=torch.channels_last)
    U = winograd.weights_forward2(torch.randn(128, 128, 3, 3, device=dev))
    with pytest.raises(Exception):
        winograd.conv3x3_U2(x, U)


def test_bev_backbone_with_winograd_matches_miopen_path(dev, monkeypatch):
    """BaseBEVBackbone with the flag (default on) against the MIOpen path (flag off): eval (BatchNorm folded, ReLU in the epilogue: one launch per layer) and the
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


def test_bev_backbone_gradients_are_as_accurate_as_the_miopen_path(dev, monkeypatch):
    """training forward + backward of the full-depth BEV backbone (5 + 5 layers, train-mode BatchNorm) on the Winograd path
    (forward, input gradient, weight gradient kernels) and on the MIOpen path, both against an f64 run of the same network:
    the backward through 11 BatchNorm layers amplifies f32 rounding to several 1e-3 on EITHER path (tools/dbg_bev_grad.py), so
    the statement that can be tested is accuracy against f64, not agreement between two f32 runs. Measured on MI355X (104 x 88,
    B = 2): output 2.3e-6 (MIOpen) / 4.0e-6 (Winograd), input gradient 3.3e-3 / 5.4e-3, weight gradients 1.0e-3 .. 4.7e-3 /
    4.3e-3 .. 6.6e-3: the Winograd transforms (differences of neighbouring pixels) cost a factor 1.3 - 4 in this amplifying
    backward (up to 7 on the last up-sampling weight, 2.1e-4 vs 1.5e-3; the MIOpen path itself moves by 3x between runs: its
    solver picks are not deterministic), both paths stay in the 1e-3 range. Bounds: output <= 1e-5, gradients <= 2e-2 (both paths)."""
    import copy
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import base_bev_backbone as bb
    from crbhip import winograd
    torch.manual_seed(3)
    cfg = EasyDict({'LAYER_NUMS': [5, 5], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [256, 256]})
    net = bb.BaseBEVBackbone(cfg, 256).to(dev).train()
    B, H, W = 2, 104, 88
    x0 = torch.randn(B, 256, H, W, device=dev) * (torch.rand(B, 1, H, W, device=dev) < 0.06)
    x0 = x0.contiguous(memory_format=torch.channels_last)
    gout = torch.randn(B, 512, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    names = ['blocks.0.1.weight', 'blocks.0.13.weight', 'blocks.1.4.weight', 'deblocks.1.0.weight']

    def run(model, x, g):
        x = x.clone().requires_grad_(True)
        y = model({'spatial_features': x})['spatial_features_2d']
        (y * g).sum().backward()
        p = dict(model.named_parameters())
        return [y.detach(), x.grad] + [p[n].grad.clone() for n in names]
    ref = run(copy.deepcopy(net).double(), x0.double(), gout.double())
    errs = {}
    for flag in (False, True):
        monkeypatch.setattr(bb, 'WINOGRAD', flag)
        monkeypatch.setattr(winograd, 'WGRAD', flag)
        got = run(copy.deepcopy(net), x0, gout)
        errs[flag] = [float((a.double() - r).norm() / r.norm()) for a, r in zip(got, ref)]
    for n, e_m, e_w in zip(['output', 'input gradient'] + names, errs[False], errs[True]):
        print('%-22s relative L2 error vs f64: MIOpen path %.2e, Winograd path %.2e' % (n, e_m, e_w))
        assert max(e_w, e_m) <= (1e-5 if n == 'output' else 2e-2), (n, e_m, e_w)


# measured on MI355X (r05, B = 8 @ 200 x 176, relative L2 against the f64 run): see the docstring; bounds = 3 x the measured error
BENCH_SHAPE_GRAD_TOL = {'output': 1e-5, 'input gradient': 2e-2, 'blocks.0.1.weight': 2e-2, 'blocks.0.13.weight': 2e-2,
                        'blocks.1.4.weight': 2e-2, 'deblocks.1.0.weight': 2e-2}


def test_bev_backbone_gradients_at_a_bench_like_shape_against_f64(dev, monkeypatch):
    """the full-depth BEV backbone (5 + 5 layers, train-mode BatchNorm) forward + backward at a bench-like
    shape - B = 8 maps of 200 x 176 with the occupancy of a KITTI BEV map, where the BatchNorm statistics are stable - on the
    Winograd path and on the MIOpen path, both against an f64 run of the same network. At B = 2 @ 104 x 88 (the test above) the
    backward amplifies f32 rounding to 3e-3 .. 7e-3 on either path; here the statistics average over 140 times more pixels. The
    bounds are 3 x the errors measured on MI355X for BOTH paths (BENCH_SHAPE_GRAD_TOL), so a regression of either path's
    gradients by more than that factor fails."""
    import copy
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import base_bev_backbone as bb
    from crbhip import winograd
    torch.manual_seed(5)
    cfg = EasyDict({'LAYER_NUMS': [5, 5], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [256, 256]})
    net = bb.BaseBEVBackbone(cfg, 256).to(dev).train()
    B, H, W = 8, 200, 176
    x0 = torch.randn(B, 256, H, W, device=dev) * (torch.rand(B, 1, H, W, device=dev) < 0.06)
    x0 = x0.contiguous(memory_format=torch.channels_last)
    gout = torch.randn(B, 512, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    names = ['blocks.0.1.weight', 'blocks.0.13.weight', 'blocks.1.4.weight', 'deblocks.1.0.weight']

    def run(model, x, g):
        x = x.clone().requires_grad_(True)
        y = model({'spatial_features': x})['spatial_features_2d']
        (y * g).sum().backward()
        p = dict(model.named_parameters())
        return [y.detach(), x.grad] + [p[n].grad.clone() for n in names]
    ref = run(copy.deepcopy(net).double(), x0.double(), gout.double())
    errs = {}
    for flag in (False, True):
        monkeypatch.setattr(bb, 'WINOGRAD', flag)
        monkeypatch.setattr(winograd, 'WGRAD', flag)
        got = run(copy.deepcopy(net), x0, gout)
        errs[flag] = [float((a.double() - r).norm() / r.norm()) for a, r in zip(got, ref)]
        del got
    for n, e_m, e_w in zip(['output', 'input gradient'] + names, errs[False], errs[True]):
        print('%-22s relative L2 error vs f64 (B=8 @200x176): MIOpen path %.2e, Winograd path %.2e' % (n, e_m, e_w))
    for n, e_m, e_w in zip(['output', 'input gradient'] + names, errs[False], errs[True]):
        assert max(e_w, e_m) <= BENCH_SHAPE_GRAD_TOL[n], (n, e_m, e_w)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['f32', 'x6'])
@pytest.mark.parametrize('shape', [(2, 64, 64, 37, 29), (3, 16, 128, 12, 21), (16, 128, 128, 40, 36), (5, 32, 64, 33, 7)])
def test_forward_kernel_writes_the_slab_sums_of_its_output(dev, monkeypatch, kernel, shape):
    """crb_conv3x3_winograd2_stats_nhwc / crb_conv3x3_winograd4_stats_nhwc: same output as the plain kernel (bit-equal), and the slab
    sums add up to the column sums of y and y^2 over the map (odd sizes: outputs of border tiles outside the map are not counted);
    bit-equal on a rerun"""
    from crbhip import winograd
    monkeypatch.setattr(winograd, 'KERNEL', kernel)
    N, C, K, H, W = shape
    torch.manual_seed(11)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
    y, st = winograd.conv3x3_stats(x, w)
    y2, st2 = winograd.conv3x3_stats(x, w)
    assert torch.equal(y, winograd.conv3x3(x, w)) and torch.equal(st, st2) and torch.equal(y, y2)
    ref1 = y.double().sum((0, 2, 3))
    ref2 = (y.double() ** 2).sum((0, 2, 3))
    got = st.double().sum(0)
    assert float((got[0] - ref1).abs().max() / ref2.sqrt().max()) < 1e-5
    assert float((got[1] - ref2).abs().max() / ref2.max()) < 1e-5


@pytest.mark.gpu
def test_bev_backbone_statistics_from_the_epilogue_match_the_statistics_pass(dev, monkeypatch):
    """BaseBEVBackbone in training with the BatchNorm statistics taken from the convolution epilogues (default) against the same
    network with the BatchNorm's own statistics pass (CRB_WINOGRAD_STATS=0): output, input gradient, a weight gradient and the
    running statistics agree to the rounding of another summation order"""
    import copy
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import base_bev_backbone as bb
    from crbhip import winograd
    torch.manual_seed(13)
    cfg = EasyDict({'LAYER_NUMS': [3, 3], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [256, 256]})
    net = bb.BaseBEVBackbone(cfg, 256).to(dev).train()
    B, H, W = 2, 52, 44
    x0 = (torch.randn(B, 256, H, W, device=dev) * (torch.rand(B, 1, H, W, device=dev) < 0.2)).contiguous(memory_format=torch.channels_last)
    gout = torch.randn(B, 512, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    outs = {}
    for flag in (False, True):
        monkeypatch.setattr(winograd, 'STATS', flag)
        model = copy.deepcopy(net)
        x = x0.clone().requires_grad_(True)
        y = model({'spatial_features': x})['spatial_features_2d']
        (y * gout).sum().backward()
        p = dict(model.named_parameters())
        outs[flag] = [y.detach(), x.grad, p['blocks.0.4.weight'].grad, model.blocks[0][5].running_mean.clone(),
                      model.blocks[1][8].running_var.clone()]
    for name, a, b, tol in zip(['output', 'input gradient', 'conv weight gradient', 'running mean', 'running var'], outs[False],
                               outs[True], [1e-5, 2e-2, 2e-2, 1e-5, 1e-5]):
        err = float((a - b).norm() / b.norm())
        assert err <= tol, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', ['f32', 'x6'])
@pytest.mark.parametrize('shape', [(4, 64, 128, 60, 44), (6, 128, 256, 100, 88)])
def test_results_do_not_depend_on_the_cu_reservation(dev, monkeypatch, kernel, shape):
    """crb_cu_reservation(n): a persistent forward launch that puts a workgroup on every CU spreads its units over (CUs - n)
    workgroups (n is clamped to half of the CUs; smaller launches ignore it); outputs are bit-equal for any n (every unit is
    computed by exactly one workgroup, whichever). First shape: 96 units (fewer than CUs: the reservation is not looked at);
    second: 836 units on 256 workgroups."""
    from crbhip import winograd, lib, check, cur_stream
    monkeypatch.setattr(winograd, 'KERNEL', kernel)
    N, C, K, H, W = shape
    torch.manual_seed(17)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)).contiguous(memory_format=torch.channels_last)
    ref = winograd.conv3x3(x, w)
    want = F.conv2d(x.double(), w.double(), padding=1)
    assert float((ref.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    try:
        for n in (16, 100, 255, 4000):
            check(lib.crb_cu_reservation(n, cur_stream(dev)), 'crb_cu_reservation')
            assert torch.equal(winograd.conv3x3(x, w), ref), n
    finally:
        check(lib.crb_cu_reservation(0, cur_stream(dev)), 'crb_cu_reservation')
    assert torch.equal(winograd.conv3x3(x, w), ref)


@pytest.mark.parametrize('kernel', ['f32', 'x6'])
def test_prepared_weight_images_equal_the_single_launches(dev, monkeypatch, kernel):
    """crb_winograd2 / 4_weights_conv_multi (all layers of a step in one launch) against one crb_winograd2 / 4_weights_conv per
    image: bit-equal, contiguous and channels_last weights, 128 / 256 channel pairs; the prepared image is handed out only for the
    same memory at the same autograd version"""
    from crbhip import winograd
    monkeypatch.setattr(winograd, 'KERNEL', kernel)
    shp = 'wino2_shape' if kernel == 'f32' else 'wino4_shape'
    torch.manual_seed(3)
    ws = [torch.randn(co, ci, 3, 3, device=dev) for co, ci in ((128, 256), (128, 128), (256, 128), (256, 256), (64, 64))]
    ws[1] = ws[1].contiguous(memory_format=torch.channels_last)
    ws += [torch.randn(128, 128, 3, 3, device=dev) for _ in range(14)]              # 19 tensors x 2 modes: two launches of <= 32 jobs
    winograd._PREPARED.clear()
    single = [(winograd.weights_forward2(w), winograd.weights_input_grad2(w)) for w in ws]
    assert winograd.prepare_weights2(ws) == 2 * len(ws)
    for w, (uf, ug) in zip(ws, single):
        pf, pg = winograd.weights_forward2(w), winograd.weights_input_grad2(w)
        assert pf is winograd._PREPARED[winograd._prep_key(w, 0)] and pg is winograd._PREPARED[winograd._prep_key(w, 1)]
        assert torch.equal(pf, uf) and torch.equal(pg, ug) and getattr(pf, shp) == getattr(uf, shp) and getattr(pg, shp) == getattr(ug, shp)
    # a weight that is freed while its image is still in the prepared set: a NEW tensor of the same shape must not be served the old
    # image (the set keeps the source alive, so the allocator cannot hand its address out again) - found by a stage-2 test that
    # failed only in the full suite order
    old = torch.randn(128, 128, 3, 3, device=dev)
    winograd.prepare_weights2([old])
    old_img = winograd.weights_forward2(old).clone()
    del old
    new = torch.randn(128, 128, 3, 3, device=dev)
    img = winograd.weights_forward2(new)
    winograd._PREPARED.clear()
    assert torch.equal(img, winograd.weights_forward2(new)) and not torch.equal(img, old_img)
    winograd.prepare_weights2(ws)
    ws[0].add_(1.0)                                                                  # an optimizer step: new version, no stale image
    fresh = winograd.weights_forward2(ws[0])
    assert fresh is not winograd._PREPARED.get(winograd._prep_key(ws[0], 0)) and not torch.equal(fresh, single[0][0])
    winograd._PREPARED.clear()


@pytest.mark.parametrize('N,C,K,H,W', [(2, 128, 128, 50, 44), (1, 256, 256, 33, 22), (3, 64, 128, 31, 9), (2, 16, 384, 33, 17),
                                       (1, 256, 128, 40, 31), (5, 32, 128, 37, 5), (3, 48, 128, 63, 70), (2, 128, 256, 47, 35)])
def test_third_form_of_the_split_bf16_kernel_is_bit_equal_to_the_second(dev, monkeypatch, N, C, K, H, W):
    """crb_conv3x3_winograd4c_nhwc (workgroup tile 32 tiles x 128 output channels, weight image in its own layout) against
    crb_conv3x3_winograd4_nhwc (64 x 64): the same arithmetic in the same order - outputs with bias + ReLU, the statistics variant's
    outputs and the column sums of its slabs (the slabs themselves cover other blocks), the input gradient: torch.equal. Ragged
    shapes: blocks that straddle two images, partial tile rows / columns, fewer units than CUs."""
    from crbhip import winograd, lib
    monkeypatch.setattr(winograd, 'KERNEL', 'x6')
    torch.manual_seed(N * 1000 + C)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 3, 3, device=dev) / np.sqrt(9 * C)
    b = torch.randn(K, device=dev)
    im = {}
    for c in (False, True):
        monkeypatch.setattr(winograd, 'FORM_C', c)
        im[c] = winograd.weights_forward4(w)
        assert bool(getattr(im[c], '_crb_c', False)) == c
    assert im[True].numel() == im[False].numel() and not torch.equal(im[True], im[False])        # same bytes, another order
    assert torch.equal(torch.sort(im[True].view(torch.int16)).values, torch.sort(im[False].view(torch.int16)).values)
    ya, yc = winograd.conv3x3_U4(x, im[False], b, relu=True), winograd.conv3x3_U4(x, im[True], b, relu=True)
    assert torch.equal(ya, yc)
    (sa_y, sa), (sc_y, sc) = winograd.conv3x3_stats_U4(x, im[False]), winograd.conv3x3_stats_U4(x, im[True])
    assert torch.equal(sa_y, sc_y) and sc.shape[0] == int(lib.crb_winograd4c_stats_slabs(N, H, W))
    ref = torch.stack([sa_y.double().sum((0, 2, 3)), (sa_y.double() ** 2).sum((0, 2, 3))])
    for s in (sa, sc):
        assert float((s.double().sum(0) - ref).abs().max() / ref.abs().max()) <= 1e-6
    if winograd.supported4(K, C, H, W) and C % 128 == 0:
        dy = torch.randn(N, K, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        g = {}
        for c in (False, True):
            monkeypatch.setattr(winograd, 'FORM_C', c)
            g[c] = winograd.conv3x3_U4(dy, winograd.weights_input_grad4(w))
        assert torch.equal(g[False], g[True])
