#!/usr/bin/env python
"""BEV backbone (KITTI config) forward + backward at B=2, 200x176: the Winograd path and the MIOpen path against an f64 run of
the same network — output, input gradient and three weight gradients as relative L2 errors. (Why: the SECOND smoke compares
sparse-layer gradients, which pass through this backbone's backward, with the CPU oracle.)"""
import copy
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from pcdet.config import EasyDict
    from pcdet.models.backbones_2d import base_bev_backbone as bb
    from crbhip import winograd
    dev = torch.device('cuda', 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.manual_seed(3)
    cfg = EasyDict({'LAYER_NUMS': [5, 5], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                    'NUM_UPSAMPLE_FILTERS': [256, 256]})
    net = bb.BaseBEVBackbone(cfg, 256).to(dev).train()
    # a sparse-looking input: 6 % of the pixels carry features (what HeightCompression hands over)
    x0 = torch.randn(B, 256, 200, 176, device=dev) * (torch.rand(B, 1, 200, 176, device=dev) < 0.06)
    x0 = x0.contiguous(memory_format=torch.channels_last)
    gout = torch.randn(B, 512, 200, 176, device=dev).contiguous(memory_format=torch.channels_last)
    names = ['blocks.0.1.weight', 'blocks.0.13.weight', 'blocks.1.4.weight', 'deblocks.1.0.weight']

    def run(model, x, g):
        x = x.clone().requires_grad_(True)
        model.zero_grad(set_to_none=True)
        y = model({'spatial_features': x})['spatial_features_2d']
        (y * g).sum().backward()
        p = dict(model.named_parameters())
        return [y.detach(), x.grad] + [p[n].grad.clone() for n in names]

    n64 = copy.deepcopy(net).double()
    ref = run(n64, x0.double(), gout.double())
    res = {}
    for tag, wino, wg, kern in (('MIOpen', False, False, 'f32'), ('Winograd f32-MFMA fwd+dgrad, MIOpen wgrad', True, False, 'f32'),
                                ('Winograd f32-MFMA fwd+dgrad+wgrad', True, True, 'f32'), ('Winograd split-bf16 fwd+dgrad, MIOpen wgrad', True, False, 'x6'),
                                ('Winograd split-bf16 fwd+dgrad, wgrad f32', True, True, 'x6'), ('MIOpen again', False, False, 'f32')):
        bb.WINOGRAD, winograd.WGRAD, winograd.KERNEL = wino, wg, kern
        got = run(copy.deepcopy(net), x0, gout)
        res[tag] = got
        errs = [float((a.double() - r).norm() / r.norm()) for a, r in zip(got, ref)]
        print('%-46s out %.2e  dx %.2e  ' % (tag, errs[0], errs[1]) + '  '.join('%s %.2e' % (n, e) for n, e in zip(names, errs[2:])), flush=True)
