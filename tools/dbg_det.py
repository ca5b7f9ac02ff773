"""Bitwise run-to-run reproducibility of the ops of the BEV training path in isolation: the hand-written row kernels and
LinearRows are reproducible, MIOpen's strided / transposed convolutions move by ~1e-7 (atomics)."""
import os, sys, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'crb-active-3ddet_amd'))
import torch
from crbhip import bnrelu
from pcdet.utils.linear_rows import LinearRows
dev = torch.device('cuda', 0)
torch.manual_seed(1)
n = 2 * 48 * 40
bns = [torch.nn.BatchNorm2d(256, eps=1e-3, momentum=0.01).to(dev).train() for _ in range(2)]
xs0 = [torch.randn(n, 256, device=dev) for _ in range(2)]
dz = torch.randn(n, 512, device=dev)
def concat_once():
    xs = [x.clone().requires_grad_(True) for x in xs0]
    out = bnrelu.bn_relu_concat(xs, bns, relu=True)
    out.backward(dz)
    return [out.detach().clone()] + [x.grad.clone() for x in xs] + [b.weight.grad.clone() for b in bns]
def check(name, fn, reps=20):
    base = fn()
    bad = 0
    for _ in range(reps):
        for b in bns:
            b.weight.grad = None; b.bias.grad = None
        r = fn()
        if not all(torch.equal(a, b) for a, b in zip(base, r)):
            bad += 1
            print('  ', name, 'differs:', [float((a - b).abs().max() / a.abs().max()) for a, b in zip(base, r)])
    print(name, 'nondeterministic runs:', bad, 'of', reps)
check('bn_relu_concat', concat_once)
w = torch.randn(256, 128, device=dev)
r0 = torch.randn(n, 128, device=dev)
dy = torch.randn(n, 256, device=dev)
def lin_once():
    r = r0.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
    LinearRows.apply(r, ww).backward(dy)
    return [r.grad.clone(), ww.grad.clone()]
check('LinearRows', lin_once)
dc = torch.nn.ConvTranspose2d(256, 256, 2, stride=2, bias=False).to(dev)
xi = torch.randn(2, 256, 24, 20, device=dev).contiguous(memory_format=torch.channels_last)
dyo = torch.randn(2, 256, 48, 40, device=dev).contiguous(memory_format=torch.channels_last)
def dc_once():
    dc.weight.grad = None
    x = xi.clone().requires_grad_(True)
    dc(x).backward(dyo)
    return [x.grad.clone(), dc.weight.grad.clone()]
check('ConvTranspose2d 2x2 s2', dc_once)
cv = torch.nn.Conv2d(128, 256, 3, stride=2, padding=1, bias=False).to(dev)
xc = torch.randn(2, 128, 48, 40, device=dev).contiguous(memory_format=torch.channels_last)
dyc = torch.randn(2, 256, 24, 20, device=dev).contiguous(memory_format=torch.channels_last)
def cv_once():
    cv.weight.grad = None
    x = xc.clone().requires_grad_(True)
    cv(x).backward(dyc)
    return [x.grad.clone(), cv.weight.grad.clone()]
check('Conv2d 3x3 s2', cv_once)
bn1 = torch.nn.BatchNorm2d(128, eps=1e-3, momentum=0.01).to(dev).train()
xb = torch.randn(n, 128, device=dev); dzb = torch.randn(n, 128, device=dev)
def bn_once():
    bn1.weight.grad = None; bn1.bias.grad = None
    x = xb.clone().requires_grad_(True)
    bnrelu.bn_relu(x, bn1, relu=True).backward(dzb)
    return [x.grad.clone(), bn1.weight.grad.clone(), bn1.bias.grad.clone()]
check('bn_relu', bn_once)
