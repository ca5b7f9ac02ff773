#!/usr/bin/env python
"""Times the training-path set-abstraction ops at the PV-RCNN RoI-grid shape (M = 16 x 128 x 216 queries against 16 x 2048
keypoints, C = 128 -> H = 64, nsample 16): gathered first layer fwd / bwd, BN+ReLU+max fwd / bwd. HIP events, median of 20."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np, torch
from crbhip import bnrelu
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U

dev = torch.device('cuda', 0)
torch.manual_seed(0)
B, NK, R, G3, C, H, ns = 16, 2048, 128, 216, 128, 64, 16
xyz = (torch.rand(B * NK, 3, device=dev) * torch.tensor([70.0, 80.0, 4.0], device=dev)).contiguous()
xc = torch.full((B,), NK, dtype=torch.int32, device=dev)
centres = xyz.view(B, NK, 3)[:, torch.randint(0, NK, (R,), device=dev)]                      # (B,R,3)
new = (centres[:, :, None, :] + (torch.rand(B, R, G3, 3, device=dev) - 0.5) * 4.0).reshape(-1, 3).contiguous()
nc = torch.full((B,), R * G3, dtype=torch.int32, device=dev)
feat = torch.randn(B * NK, C, device=dev, requires_grad=True)
w = (torch.randn(H, 3 + C, device=dev) * 0.1).requires_grad_(True)
ball = U.ball_query(1.6, ns, xyz, xc, new, nc)
print('empty fraction %.3f' % float(ball[1].float().mean()))


def timeit(fn, n=20):
    for _ in range(3): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))


y = U.grouped_first_layer_rows(1.6, ns, xyz, xc, new, nc, feat, w, ball=ball)
go = torch.randn_like(y)
print('first layer fwd   %.3f ms' % timeit(lambda: U.grouped_first_layer_rows(1.6, ns, xyz, xc, new, nc, feat, w, ball=ball)))
print('first layer bwd   %.3f ms' % timeit(lambda: torch.autograd.grad(y, (feat, w), go, retain_graph=True)))
bn = torch.nn.BatchNorm1d(H).to(dev).train()
x = y.detach().requires_grad_(True)
z = bnrelu.bn_relu_max_concat([x], [ns], [bn])
gz = torch.randn_like(z)
print('bn+relu+max fwd   %.3f ms' % timeit(lambda: bnrelu.bn_relu_max_concat([x], [ns], [bn])))
print('bn+relu+max bwd   %.3f ms' % timeit(lambda: torch.autograd.grad(z, x, gz, retain_graph=True)))
z1 = bnrelu.bn_relu(x, bn)
g1 = torch.randn_like(z1)
print('bn+relu fwd       %.3f ms' % timeit(lambda: bnrelu.bn_relu(x, bn)))
print('bn+relu bwd       %.3f ms' % timeit(lambda: torch.autograd.grad(z1, x, g1, retain_graph=True)))
print('rows: %d x %d = %.2f GB' % (y.shape[0], H, y.numel() * 4 / 1e9))

from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        torch.autograd.grad(y, (feat, w), go, retain_graph=True)
    torch.cuda.synchronize()
print('first layer bwd, kernels (us per call):')
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print('  %9.1f  x%d  %s' % (e.device_time_total / 5, e.count // 5, e.key[:110]))
