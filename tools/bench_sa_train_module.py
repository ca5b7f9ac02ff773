#!/usr/bin/env python
"""StackSAModuleMSG in TRAINING mode at the PV-RCNN RoI-grid shape (M = 16 x 128 x 216 queries against 16 x 2048 keypoints,
C = 128, MLPS [[64, 64], [64, 64]], nsample 16; pvrcnn_head.py:102-113): the recompute node (csrc/sa_mlp_train.hip) against the
rows path (CRB_SA_TRAIN_FUSED=0): forward / forward+backward time (HIP events, median), peak memory of a forward+backward, and the
kernels of one forward+backward of each. usage: python tools/bench_sa_train_module.py [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pcdet.config import EasyDict  # noqa: E402
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as M  # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
if os.environ.get('SA_BENCH', 'roi') == 'vsa':
    # a voxel level of VoxelSetAbstraction (x_conv3-like): 6,500 source voxels per frame on a thin slab (lidar surfaces), 2,048
    # keypoints per frame drawn from them, radii 1.2 / 2.4 m, nsample 16 / 32, 64 channels: every ball full of different rows
    NK, R, G3, C = 6500, 2048, 1, 64
    xyz = (torch.rand(B * NK, 3, device=dev) * torch.tensor([70.0, 80.0, 1.0], device=dev)).contiguous()
    xc = torch.full((B,), NK, dtype=torch.int32, device=dev)
    new = xyz.view(B, NK, 3)[:, torch.randperm(NK, device=dev)[:R]].reshape(-1, 3).contiguous()
    nc = torch.full((B,), R, dtype=torch.int32, device=dev)
    feat = torch.randn(B * NK, C, device=dev, requires_grad=True)
    layer, c_out = M.build_local_aggregation_module(C, EasyDict({'MLPS': [[64, 64], [64, 64]], 'POOL_RADIUS': [1.2, 2.4],
                                                                'NSAMPLE': [16, 32]}))
else:
    NK, R, G3, C = 2048, 128, 216, 128
    xyz = (torch.rand(B * NK, 3, device=dev) * torch.tensor([70.0, 80.0, 4.0], device=dev)).contiguous()
    xc = torch.full((B,), NK, dtype=torch.int32, device=dev)
    centres = xyz.view(B, NK, 3)[:, torch.randint(0, NK, (R,), device=dev)]
    new = (centres[:, :, None, :] + (torch.rand(B, R, G3, 3, device=dev) - 0.5) * 4.0).reshape(-1, 3).contiguous()
    nc = torch.full((B,), R * G3, dtype=torch.int32, device=dev)
    feat = torch.randn(B * NK, C, device=dev, requires_grad=True)
    layer, c_out = M.build_local_aggregation_module(C, EasyDict({'MLPS': [[64, 64], [64, 64]], 'POOL_RADIUS': [0.8, 1.6],
                                                                'NSAMPLE': [16, 16]}))
layer = layer.to(dev).train()
go = torch.randn(new.shape[0], c_out, device=dev)
params = [feat] + list(layer.parameters())


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def fwd():
    return layer(xyz, xc, new, nc, feat, query_group=G3 if G3 > 1 else None)[1]


def fwd_bwd():
    out = fwd()
    return torch.autograd.grad(out, params, go)


if os.environ.get('CRB_MEASURE_LIB') == '1' and os.environ.get('SAT_SKIP'):
    # skip-work builds of the recompute passes (measurement library): where a pass spends its time
    from crbhip import lib
    for bits in [int(v) for v in os.environ['SAT_SKIP'].split(',')]:
        lib.crb_sa_mlp2_train_set_skip(bits)
        M.FUSED_TRAIN = True
        with torch.no_grad():
            pass
        from torch.profiler import profile, ProfilerActivity
        fwd_bwd()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fwd_bwd()
            torch.cuda.synchronize()
        evs = sorted([e for e in prof.events() if e.device_time_total > 0 and 'sa_train' in e.name], key=lambda e: e.time_range.start)
        print('skip bits %d (1 = no forward-GEMM MFMAs, 2 = P row 0 for every sample, 4 = no first layer): sa_train launches in order (us): %s'
              % (bits, ', '.join('%.0f' % e.device_time_total for e in evs)))
    lib.crb_sa_mlp2_train_set_skip(0)
    sys.exit(0)
res = {}
for tag, flag in (('recompute node', True), ('rows path', False)):
    M.FUSED_TRAIN = flag
    t_f = timeit(fwd)
    t_fb = timeit(fwd_bwd)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    g = fwd_bwd()
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    res[tag] = (fwd().detach(), [t.detach() for t in g])
    print('%-15s forward %.3f ms, forward+backward %.3f ms, peak memory above the inputs %.2f GB' % (tag, t_f, t_fb, peak / 2**30))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize()
    tot = sum(e.device_time_total for e in prof.key_averages()) / 3
    print('   kernels of one forward+backward: %.3f ms in %d launches' % (tot / 1e3, sum(e.count for e in prof.key_averages()) // 3))
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]:
        print('   %9.1f us  x%-3d %s' % (e.device_time_total / 3, e.count // 3, e.key[:120]))
M.FUSED_TRAIN = True
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    fwd_bwd()
    torch.cuda.synchronize()
print('recompute node, launches of one forward+backward in order (us):')
evs = sorted([e for e in prof.events() if e.device_time_total > 0 and ('sa_train' in e.name or 'group_affine' in e.name or 'Radix' in e.name or 'radix' in e.name or 'pair_source' in e.name)], key=lambda e: e.time_range.start)
print('   ' + ', '.join('%s %.0f' % (e.name.split('(')[0].split('::')[-1][:34], e.device_time_total) for e in evs))
print('empty balls: %s of the queries at r = 0.8 / 1.6' % ', '.join('%.3f' % float(bl[1].float().mean()) for bl in layer._balls(xyz, xc, new, nc, G3 if G3 > 1 else None)))
a, b = res['recompute node'], res['rows path']
print('recompute vs rows: output max |diff| %.2e (scale %.2e)' % (float((a[0] - b[0]).abs().max()), float(b[0].abs().max())))
for k, (x, y) in enumerate(zip(a[1], b[1])):
    print('   grad %d %s: rel L2 %.2e' % (k, tuple(x.shape), float((x - y).norm() / y.norm().clamp_min(1e-20))))
print('empty balls: %.3f of the queries at r = 0.8' % float(layer._balls(xyz, xc, new, nc, G3 if G3 > 1 else None)[0][1].float().mean()))
