#!/usr/bin/env python
"""Every output of the sa_mlp_train.hip entry points on a small two-frame case against a float64 torch restatement of the same
algebra (pointnet2_modules.py:90-108 in training mode), one entry point at a time. usage: python tools/dbg_sa_train.py [h1 h2 ns]"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from crbhip import lib, check, ptr, cur_stream, bnrelu  # noqa: E402
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U  # noqa: E402
from synth import kitti_batch  # noqa: E402

h1, h2, ns = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 16)
dev = torch.device('cuda', 0)
torch.manual_seed(1)
pts, off, _ = kitti_batch(2, 7, n_points=6000)
xyz = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).to(dev)
xc = torch.from_numpy(np.diff(off).astype(np.int32)).to(dev)
rng = np.random.default_rng(2)
sel = np.concatenate([rng.choice(6000, 500, replace=False), 6000 + rng.choice(6000, 301, replace=False)])
new = xyz[torch.from_numpy(sel).to(dev)].contiguous()
new[::3] += 55.0
nc = torch.tensor([500, 301], dtype=torch.int32, device=dev)
C = 20
feat = torch.randn(12000, C, device=dev)
W1 = torch.randn(h1, 3 + C, device=dev) * 0.3
W2 = torch.randn(h2, h1, device=dev) * 0.3
g1, b1 = torch.randn(h1, device=dev) * 0.5 + 0.8, torch.randn(h1, device=dev) * 0.3
g2, b2 = torch.randn(h2, device=dev) * 0.5 + 0.8, torch.randn(h2, device=dev) * 0.3
idx, empty = U.ball_query(1.2, ns, xyz, xc, new, nc)
M, B = new.shape[0], 2
n = M * ns
st = cur_stream(dev)
print('M %d, ns %d, h %d -> %d, empty %.2f' % (M, ns, h1, h2, float(empty.float().mean())))

# ---- float64 restatement
start = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), xc.long().cumsum(0)[:-1]])
qframe = torch.repeat_interleave(torch.arange(B, device=dev), nc.long())
rows = start[qframe][:, None] + idx.long()                              # (M, ns)
em = empty.bool()
rel = (xyz[rows] - new[:, None, :]).double()
grp = torch.cat([rel, feat[rows].double()], 2)
grp[em] = 0
x = grp.reshape(n, 3 + C)
y1 = x @ W1.double().t()
mu1, var1 = y1.mean(0), y1.var(0, unbiased=False)
is1 = (var1 + 1e-5).rsqrt()
xh1 = (y1 - mu1) * is1
z1 = (g1.double() * xh1 + b1.double()).clamp_min(0)
y2 = z1 @ W2.double().t()
mu2, var2 = y2.mean(0), y2.var(0, unbiased=False)
is2 = (var2 + 1e-5).rsqrt()
xh2 = (y2 - mu2) * is2
z2 = (g2.double() * xh2 + b2.double()).clamp_min(0)
out_ref, arg_ref = z2.view(M, ns, h2).max(1)


def rel_err(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# ---- forward entry points
xcn, ncn = xc.contiguous(), nc.contiguous()
w1x = W1[:, :3].t().contiguous()
w1f = W1[:, 3:].contiguous()
P = feat @ w1f.t()
emu = empty.to(torch.uint8).contiguous()
nslab = int(lib.crb_group_affine_rows_grad_blocks(M, ns))
slab1 = torch.empty((nslab, 2, h1), device=dev)
check(lib.crb_group_affine_rows_stats_stack(B, M, h1, ns, ptr(xyz), ptr(xcn), ptr(P), ptr(new), ptr(ncn), ptr(idx), ptr(emu), ptr(w1x),
                                            None, None, ptr(slab1), st), 'stats0')
s = slab1.double().sum(0)
print('pass 0: mean1 %.2e, E[y1^2] %.2e' % (rel_err(s[0] / n, mu1), rel_err(s[1] / n, (y1 * y1).mean(0))))
mean1, invstd1 = mu1.float().contiguous(), is1.float().contiguous()
nwave = int(lib.crb_sa_mlp2_train_waves(M))
slab2 = torch.empty((nwave, 2, h2), device=dev)
check(lib.crb_sa_mlp2_train_stats(B, M, ns, h1, h2, ptr(xyz), ptr(xcn), ptr(P), ptr(new), ptr(ncn), ptr(idx), ptr(emu), ptr(w1x),
                                  ptr(mean1), ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(slab2), st), 'statsA')
s = slab2.double().sum(0)
print('pass A: mean2 %.2e, E[y2^2] %.2e' % (rel_err(s[0] / n, mu2), rel_err(s[1] / n, (y2 * y2).mean(0))))
mean2, invstd2 = mu2.float().contiguous(), is2.float().contiguous()
out = torch.empty((M, h2 + 8), device=dev)
arg = torch.empty((M, h2), dtype=torch.int32, device=dev)
ysel = torch.empty((M, h2), device=dev)
check(lib.crb_sa_mlp2_train_max(B, M, ns, h1, h2, ptr(xyz), ptr(xcn), ptr(P), ptr(new), ptr(ncn), ptr(idx), ptr(emu), ptr(w1x), ptr(mean1),
                                ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2),
                                ctypes.c_void_p(out.data_ptr() + 16), h2 + 8, ptr(arg), ptr(ysel), st), 'max')
o = out[:, 4:4 + h2]
ysel_ref = y2.view(M, ns, h2).gather(1, arg.long()[:, None, :])[:, 0]
z_at_arg = z2.view(M, ns, h2).gather(1, arg.long()[:, None, :])[:, 0]
print('pass B: out %.2e; z at the kernel\'s arg vs the max %.2e; arg == torch arg on %.4f; ysel %.2e (live %.2e, empty %.2e)' % (
    rel_err(o, out_ref), rel_err(z_at_arg, out_ref), float((arg.long() == arg_ref).float().mean()), rel_err(ysel, ysel_ref),
    rel_err(ysel[~em], ysel_ref[~em]), rel_err(ysel[em], ysel_ref[em])))

# ---- backward
go = torch.randn(M, h2 + 8, device=dev)
gsel = go[:, 4:4 + h2].double()
dz2 = torch.zeros(M, ns, h2, dtype=torch.float64, device=dev)
dz2.scatter_(1, arg.long()[:, None, :], (gsel * (z_at_arg > 0))[:, None, :])
dz2 = dz2.view(n, h2)
db2, dg2 = dz2.sum(0), (dz2 * xh2).sum(0)
dy2 = g2.double() * is2 * (dz2 - db2 / n - xh2 * dg2 / n)
dz1 = (dy2 @ W2.double()) * (z1 > 0)
dW2_ref = dy2.t() @ z1
db1, dg1 = dz1.sum(0), (dz1 * xh1).sum(0)
dy1 = g1.double() * is1 * (dz1 - db1 / n - xh1 * dg1 / n)
dW1_ref = dy1.t() @ x
gx = dy1 @ W1.double()
gfeat_ref = torch.zeros(12000, C, dtype=torch.float64, device=dev)
live_rows = (~em)[:, None].expand(M, ns).reshape(-1)
gfeat_ref.index_add_(0, rows.reshape(-1)[live_rows], gx[live_rows][:, 3:])

d2 = torch.empty((2, h2), device=dev)
wsb = lib.crb_bn_workspace_bytes(M, h2)
ws, tk = bnrelu._scratch(dev, wsb)
gp = ctypes.c_void_p(go.data_ptr() + 16)
check(lib.crb_bn_relu_max_backward_sums(ptr(ysel), gp, h2 + 8, M, h2, ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2), ptr(d2[1]), ptr(d2[0]),
                                        ptr(ws), wsb, ptr(tk), st), 'sums')
print('sums: dbeta2 %.2e, dgamma2 %.2e' % (rel_err(d2[0], db2), rel_err(d2[1], dg2)))
gz1 = torch.full((n, h1), float('nan'), device=dev)
d1 = torch.empty((2, h1), device=dev)
dW2 = torch.empty((h2, h1), device=dev)
wsf = int(lib.crb_sa_mlp2_train_backward_workspace_floats(M, h1, h2))
wsp = torch.empty((wsf,), device=dev)
db2f, dg2f = db2.float().contiguous(), dg2.float().contiguous()
check(lib.crb_sa_mlp2_train_backward(B, M, ns, h1, h2, ptr(xyz), ptr(xcn), ptr(P), ptr(new), ptr(ncn), ptr(idx), ptr(emu), ptr(w1x), ptr(mean1),
                                     ptr(invstd1), ptr(g1), ptr(b1), ptr(W2), ptr(mean2), ptr(invstd2), ptr(g2), ptr(b2), gp, h2 + 8,
                                     ptr(arg), ptr(db2f), ptr(dg2f), ptr(gz1), ptr(d1), ptr(dW2), ptr(wsp), wsf, st), 'bwd')
gz1v = gz1.view(M, ns, h1)
print('pass C: gz1 (live rows) %.2e, rows of empty balls untouched %s; dbeta1 %.2e, dgamma1 %.2e, dW2 %.2e' % (
    rel_err(gz1v[~em], dz1.view(M, ns, h1)[~em]), bool(torch.isnan(gz1v[em]).all()), rel_err(d1[0], db1), rel_err(d1[1], dg1),
    rel_err(dW2, dW2_ref)))
gP = torch.zeros((12000, h1), device=dev)
part = torch.empty((nslab, 3, h1), device=dev)
db1f, dg1f = db1.float().contiguous(), dg1.float().contiguous()
check(lib.crb_group_affine_rows_grad_bn_recompute_stack(B, M, h1, ns, ptr(xyz), ptr(xcn), ptr(P), ptr(new), ptr(ncn), ptr(idx), ptr(emu),
                                                        ptr(w1x), ptr(gz1), ptr(mean1), ptr(invstd1), ptr(g1), ptr(b1), ptr(db1f),
                                                        ptr(dg1f), None, None, 0, ptr(gP), ptr(part), st), 'passD')
gW1 = torch.cat([part.sum(0).t(), gP.t() @ feat], 1)
print('pass D: dW1 %.2e, dfeatures %.2e' % (rel_err(gW1, dW1_ref), rel_err(gP @ w1f, gfeat_ref)))
