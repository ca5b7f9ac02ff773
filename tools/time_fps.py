"""GPU, measurement library: fps_kernel (round 2) against fps2_kernel (round 6) at the PV-RCNN size, picks compared."""
import os
import sys
os.environ['CRB_MEASURE_LIB'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from crbhip import lib  # noqa: E402
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pu  # noqa: E402
from synth import kitti_batch  # noqa: E402

dev = torch.device('cuda:0')
for B, n, m in ((16, 20000, 2048), (64, 20000, 2048), (4, 4096, 512), (2, 8000, 1024)):
    pts, off, _ = kitti_batch(0, B, n)
    xyz = torch.from_numpy(pts[:, :3].reshape(B, n, 3).copy()).to(dev)
    res = {}
    for variant in (1, 2, 1, 2):
        lib.crb_fps_set_variant(variant)
        out = pu.farthest_point_sample(xyz, m)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = pu.farthest_point_sample(xyz, m)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res[variant] = out.clone()
        print('B=%d n=%d m=%d variant %d: %.3f ms (%.2f us per round)' % (B, n, m, variant, ms, 1e3 * ms / (m - 1)), flush=True)
    print('   picks equal: %s' % bool(torch.equal(res[1], res[2])), flush=True)
lib.crb_fps_set_variant(2)
