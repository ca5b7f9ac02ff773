#!/usr/bin/env python
"""Per-workgroup lifetime / placement of the v2 wgrad kernel (measurement build hook crb_sparse_conv_set_wgrad_debug):
how many workgroups each CU got, when they started and ended (wall_clock64 ticks = 10 ns), steps per workgroup.
Usage: python tools/wgrad_timeline.py [level]"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import collections  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import lib, sparse, voxel
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lib.crb_sparse_conv_set_wgrad_mode(mode)
    dev = torch.device('cuda', 0)
    pts, off, _ = kitti_batch(0, 16)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), KITTI_RANGE, KITTI_VOXEL, 16000, 5,
                       want_voxels=False, want_mean=True)
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    chans = {1: (16, 16), 2: (32, 32), 3: (64, 64), 4: (64, 64)}
    for lvl in range(2, level + 1):
        rbs = sparse.spconv_rulebook(coords, shape, 16, *geo[lvl - 2])
        coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
    rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
    cin, cout = chans[level]
    n = rb.n_out
    x, dy = torch.randn(n, cin, device=dev), torch.randn(n, cout, device=dev)
    pairs = rb.pairs()
    for _ in range(3):
        sparse._conv_wgrad_raw(x, dy, pairs, 27)
    dbg = torch.zeros((8192, 4), dtype=torch.int64, device=dev)
    lib.crb_sparse_conv_set_wgrad_debug(dbg.data_ptr())
    sparse._conv_wgrad_raw(x, dy, pairs, 27)
    torch.cuda.synchronize()
    lib.crb_sparse_conv_set_wgrad_debug(None)
    d = dbg.cpu().numpy()
    d = d[d[:, 1] != 0]
    t0 = d[:, 0].min()
    start, end = d[:, 0] - t0, d[:, 1] - t0
    hw = d[:, 2]
    xcc = d[:, 3] & 0xffffffff
    steps = d[:, 3] >> 32
    loop_raw = hw >> 32
    hw = hw & 0xffffffff
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu
    per_cu = collections.Counter(key.tolist())
    print('workgroups %d, distinct CUs %d, wg per CU histogram %s' % (len(d), len(per_cu), sorted(collections.Counter(per_cu.values()).items())))
    print('ticks: kernel span %d, wg lifetime median %d (p10 %d, p90 %d); start median %d p90 %d max %d' % (
        end.max(), np.median(end - start), np.percentile(end - start, 10), np.percentile(end - start, 90), np.median(start),
        np.percentile(start, 90), start.max()))
    print('steps per wg: min %d median %d max %d' % (steps.min(), np.median(steps), steps.max()))
    loop = loop_raw.astype(np.float64)
    full = steps == steps.max()
    print('mode %d: shader cycles per step of wave 0 (full workgroups): median %.0f p10 %.0f p90 %.0f' % (
        mode, np.median(loop[full] / steps[full]), np.percentile(loop[full] / steps[full], 10), np.percentile(loop[full] / steps[full], 90)))
    late = start > 0.2 * end.max()
    print('workgroups starting after 20%% of the span: %d' % late.sum())
    # concurrency over time
    ev = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
    cur, last, area = 0, 0, collections.Counter()
    for t, dlt in ev:
        area[cur] += t - last
        last, cur = t, cur + dlt
    tot = sum(area.values())
    print('resident workgroups over time: ' + ', '.join('%d: %.0f%%' % (k, 100.0 * v / tot) for k, v in sorted(area.items()) if v / tot > 0.02))
