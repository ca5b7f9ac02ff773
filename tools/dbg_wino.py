"""Run-to-run differences of the BaseBEVBackbone training gradients at a small shape, default path (flag False) and Winograd
path (flag True) alternating, against the first default run: both show the same bimodal 2e-2 deviations (see
tests/test_winograd_gpu.py)."""
import os, sys, copy
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'crb-active-3ddet_amd'))
import torch
from pcdet.config import EasyDict
from pcdet.models.backbones_2d import base_bev_backbone as bb
dev = torch.device('cuda', 0)
torch.manual_seed(3)
cfg = EasyDict({'LAYER_NUMS': [2, 2], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2], 'NUM_UPSAMPLE_FILTERS': [256, 256]})
net = bb.BaseBEVBackbone(cfg, 256).to(dev)
x = torch.randn(2, 256, 48, 40, device=dev).contiguous(memory_format=torch.channels_last)
def run(flag):
    bb.WINOGRAD = flag
    n2 = copy.deepcopy(net); n2.train()
    xg = x.clone().requires_grad_(True)
    tr = n2({'spatial_features': xg})['spatial_features_2d']
    tr.square().mean().backward()
    g = {k: p.grad.clone() for k, p in n2.named_parameters()}
    g['INPUT'] = xg.grad.clone(); g['OUT'] = tr.detach().clone()
    return g
base = run(False)
for it in range(10):
    g = run(it % 2 == 1)
    print('flag', it % 2 == 1, end=' ')
    worst = sorted(((float((g[k]-base[k]).abs().max()/base[k].abs().max()), k) for k in g), reverse=True)[:4]
    print(it, ['%s %.1e' % (k, e) for e, k in worst], flush=True)
