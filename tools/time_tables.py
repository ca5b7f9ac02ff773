#!/usr/bin/env python
"""Where the table building of one SECOND bs=16 step goes: whole plan (host wall incl. the read-back, and device time by
events), then the finish pass alone — all 12 tables in two launches — in its measurement builds (no sort / no packed fill / no
pair lists; wrong tables by design, libcrbhip_measure.so).  usage: python tools/time_tables.py [kitti|waymo] [batch]"""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

SPECS = [('subm', (3, 3, 3)), ('spconv', (3, 3, 3), (2, 2, 2), (1, 1, 1)), ('subm', (3, 3, 3)),
         ('spconv', (3, 3, 3), (2, 2, 2), (1, 1, 1)), ('subm', (3, 3, 3)), ('spconv', (3, 3, 3), (2, 2, 2), (0, 1, 1)),
         ('subm', (3, 3, 3)), ('spconv', (3, 1, 1), (2, 1, 1), (0, 0, 0))]


def main():
    from crbhip import lib, sparse, voxel
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL, WAYMO_RANGE, WAYMO_VOXEL
    kind = sys.argv[1] if len(sys.argv) > 1 else 'kitti'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if kind == 'kitti' else 4)
    dev = torch.device('cuda', 0)
    waymo = kind == 'waymo'
    pts, off, _ = kitti_batch(0, B, 160000 if waymo else 20000, waymo=waymo)
    rng_, vs = (WAYMO_RANGE, WAYMO_VOXEL) if waymo else (KITTI_RANGE, KITTI_VOXEL)
    r = voxel.voxelize(torch.from_numpy(pts).to(dev), torch.from_numpy(off).to(dev), rng_, vs, 150000 if waymo else 16000, 5,
                       want_voxels=False, want_mean=True)
    coords = r['coords']
    shape = [41, 1504, 1504] if waymo else [41, 1600, 1408]

    def plan():
        with torch.enable_grad():
            return sparse.build_rulebooks(coords, shape, B, SPECS, want_grad=True)
    for _ in range(5):
        books = plan()
    torch.cuda.synchronize()
    ws, ds = [], []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        e0.record()
        plan()
        e1.record()
        torch.cuda.synchronize()
        ws.append((time.perf_counter() - t) * 1e6)
        ds.append(e0.elapsed_time(e1) * 1e3)
    print('%s bs=%d: rows per level %s' % (kind, B, [b.n_out for b in books]))
    print('whole plan (8 rulebooks, 12 tables, pairs): host wall %.0f us (median), first-to-last kernel %.0f us' % (
        np.median(ws), np.median(ds)))
    tabs, flags = [], []
    for b in books:
        tabs.append(b.table('nbr'))
        flags.append(True)
        if not b.subm:
            tabs.append(b.table('nbr_t'))
            flags.append(False)
    for skip, what in ((0, 'normal'), (1, 'no sort'), (2, 'no packed fill'), (4, 'no pair lists'), (6, 'sort only'), (7, 'keys + prefix only')):
        lib.crb_tables_set_skip(skip)
        plans, keep = sparse.plan_tables(tabs, flags)
        st = sparse.cur_stream(dev)
        for _ in range(3):
            lib.crb_tables_finish(plans, len(tabs), st)
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            lib.crb_tables_finish(plans, len(tabs), st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print('finish pass, %-20s %.0f us (median of 10, the three launches only)' % (what + ':', np.median(ts)))
    lib.crb_tables_set_skip(0)


if __name__ == '__main__':
    main()
