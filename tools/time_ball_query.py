"""GPU: the set-abstraction ball queries on the per-call cell grid (crb_ball_query2_grid_stack) against the scans
(crb_ball_query2_stack) at the PV-RCNN shapes: 2,048 keypoints per frame against raw points / voxel centres of the four levels."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U  # noqa: E402
from synth import kitti_batch  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (16, 64):
    pts, off, _ = kitti_batch(0, B, 20000)
    raw = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).to(dev)
    rc = torch.full((B,), 20000, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(0)
    new = torch.cat([raw[b * 20000:(b + 1) * 20000][torch.from_numpy(rng.choice(20000, 2048, replace=False)).to(dev)] for b in range(B)]).contiguous()
    nc = torch.full((B,), 2048, dtype=torch.int32, device=dev)
    # voxel-centre sources: the occupied voxels of the frames at strides 1, 2, 4, 8 (0.05 x 0.05 x 0.1 m voxels)
    srcs = [('raw points', raw, rc, (0.4, 16, 0.8, 16))]
    for name, stride, radii in (('x_conv1', 1, (0.4, 16, 0.8, 16)), ('x_conv2', 2, (0.8, 16, 1.2, 32)), ('x_conv3', 4, (1.2, 16, 2.4, 32)),
                                ('x_conv4', 8, (2.4, 16, 4.8, 32))):
        vs = np.array([0.05, 0.05, 0.1]) * stride
        cs, cnt = [], []
        for b in range(B):
            p = pts[off[b]:off[b + 1], :3]
            ijk = np.unique(np.floor((p - np.array([0, -40, -3])) / vs).astype(np.int64), axis=0)
            cs.append((ijk + 0.5) * vs + np.array([0, -40, -3]))
            cnt.append(len(ijk))
        srcs.append((name, torch.from_numpy(np.concatenate(cs).astype(np.float32)).to(dev), torch.tensor(cnt, dtype=torch.int32, device=dev), radii))
    for name, xyz, xc, (ra, na, rb, nb) in srcs:
        res = {}
        for grid in (False, True, False, True):
            U.BALL_QUERY_GRID = grid
            t = timeit(lambda: U.ball_query_pair(ra, na, rb, nb, xyz, xc, new, nc))
            res[grid] = U.ball_query_pair(ra, na, rb, nb, xyz, xc, new, nc)
            print('B=%d %-10s n=%7d radii %.1f/%.1f: %s %.1f us' % (B, name, xyz.shape[0], ra, rb, 'grid' if grid else 'scan', t), flush=True)
        same = all(torch.equal(a, b) for pa, pb in zip(res[False], res[True]) for a, b in zip(pa, pb))
        print('    lists equal: %s' % same, flush=True)
U.BALL_QUERY_GRID = True
