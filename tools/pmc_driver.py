#!/usr/bin/env python
"""Minimal driver for PMC passes: build the SECOND bs=16 level-3 (64->64 subm) rulebook and launch the gather-GEMM forward
a few times. Kept tiny so that a counter-collection pass (which serialises every kernel) finishes in seconds."""
import os
os.environ.setdefault('CRB_MEASURE_LIB', '1')     # measurement build of the library (include/crb_hip_measure.h)
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
import torch  # noqa: E402

if __name__ == '__main__':
    from crbhip import sparse, voxel
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    kind = sys.argv[3] if len(sys.argv) > 3 else 'fwd'            # fwd | wgrad | bf16x3
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
    P = int((rb.nbr >= 0).sum())
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) / 10
    table = rb.table_for('nbr', cin, cout, 'bf16x3' if kind == 'bf16x3' else 'f32') if not os.environ.get('CRB_NO_COMPACT') else rb.sorted_table('nbr')
    dy = torch.randn(n, cout, device=dev)
    pairs = rb.pairs()
    torch.cuda.synchronize()
    if kind == 'bf16x3':
        from crbhip import lib
        lib.crb_sparse_conv_bf16x3_set_tiles_per_wave(int(os.environ.get('CRB_BF16X3_TPW', '0')))
        lib.crb_sparse_conv_bf16x3_set_mode(int(os.environ.get('CRB_BF16X3_MODE', '0')))
    for _ in range(iters):
        if kind == 'wgrad':
            sparse._conv_wgrad_raw(x, dy, pairs, 27)
        else:
            sparse._conv_forward_raw(x, w, table, n, arithmetic='bf16x3' if kind == 'bf16x3' else 'f32')
    torch.cuda.synchronize()
    balg = 4.0 * n * cin + 4.0 * n * cout + 8.0 * P + 4.0 * 27 * cin * cout
    print('PMC_DRIVER %s level %d N=%d P=%d alg_bytes=%d flops=%d' % (kind, level, n, P, balg, 2 * P * cin * cout))
