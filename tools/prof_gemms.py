#!/usr/bin/env python
"""Every dense GEMM of one PV-RCNN (or SECOND) training step with its shape, device time and fraction of the f32 MFMA peak:
torch.profiler with record_shapes over aten::mm / addmm / bmm / linear / convolution (1x1 and conv1d run as GEMMs); device time =
the kernels launched by the op. usage: python tools/prof_gemms.py [pvrcnn|second]"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'pvrcnn'
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import pv_rcnn_cfg, second_cfg
    from pcdet.models import build_network
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_network((pv_rcnn_cfg() if which == 'pvrcnn' else second_cfg()).MODEL, 3, SyntheticDataset(num_frames=2)).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    pts, off, gt = kitti_batch(100, 16, 20000)
    bidx = np.repeat(np.arange(16, dtype=np.float32), np.diff(off))
    batch = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
             'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
             'batch_size': 16, 'point_frame_counts_host': np.diff(off).tolist()}

    def step():
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = model(dict(batch))
        ret['loss'].backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    names = ('aten::mm', 'aten::addmm', 'aten::bmm', 'aten::baddbmm')
    rows = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for e in prof.events():
        if e.name not in names:
            continue
        ks = list(getattr(e, 'kernels', []))
        if not ks:
            continue
        shp = [tuple(s) for s in (e.input_shapes or []) if s]
        if e.name in ('aten::mm',):
            (m, k), (_, n) = shp[0], shp[1]
            bsz = 1
        elif e.name == 'aten::addmm':
            (m, k), (_, n) = shp[1], shp[2]
            bsz = 1
        else:
            a, b = (shp[0], shp[1]) if e.name == 'aten::bmm' else (shp[1], shp[2])
            bsz, m, k, n = a[0], a[1], a[2], b[2]
        key = (e.name.replace('aten::', ''), bsz, m, n, k)
        rows[key][0] += 1
        rows[key][1] += sum(kk.duration for kk in ks)
        for kk in ks:
            rows[key][2][kk.name[:60]] += 1
    tot = sum(v[1] for v in rows.values())
    print('%s step: %d GEMM calls, %.2f ms of device time' % (which, sum(v[0] for v in rows.values()), tot / 1e3))
    print('   us total  calls  op      batch x M x N x K                GFLOP   TF/s  of 157.3   kernel')
    for key, (n, us, ker) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
        op, bsz, m, nn, k = key
        gf = 2.0 * bsz * m * nn * k * n / 1e9
        print('  %8.1f  %4d   %-6s %3d x %7d x %5d x %6d  %7.2f  %6.1f  %5.2f   %s' % (us, n, op, bsz, m, nn, k, gf, gf / us * 1e3 if us else 0,
                                                                                   gf / us * 1e3 / 157.3 if us else 0, ker.most_common(1)[0][0]))
