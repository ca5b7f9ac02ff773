#!/usr/bin/env python
"""Per-kernel roofline table at BASELINE size (16 synthetic KITTI frames x 20k points): every hand-written op of SURVEY
§8(a) timed through its Python wrapper (C-ABI launches only, HIP events, median of N), with the algorithmic bytes / flops
of SURVEY §8(d) and the fraction of the binding roof (HBM 8 TB/s, f32 MFMA 157.3 TF). Prints a markdown table.
Usage: python tools/bench_kernels.py [--iters 30] > profiles/rNN_kernel_rooflines.md"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM, MFMA = 8.0e12, 157.3e12
ROWS = []


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


def row(name, us, nbytes=None, flops=None, note=''):
    gbs = nbytes / us / 1e3 if nbytes else None
    tf = flops / us / 1e6 if flops else None
    if nbytes and flops and flops / MFMA > nbytes / HBM:
        bound, frac = 'mfma f32', tf * 1e12 / MFMA
    elif nbytes:
        bound, frac = 'hbm', gbs * 1e9 / HBM
    else:
        bound, frac = 'latency', None
    ROWS.append('| %s | %.1f | %s | %s | %s | %s | %s |' % (
        name, us, '%.1f MB' % (nbytes / 1e6) if nbytes else '—', '%.0f GB/s' % gbs if gbs else '—',
        '%.1f TF' % tf if tf else '—', bound + (' %.1f %%' % (100 * frac) if frac is not None else ''), note))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=30)
    a = ap.parse_args()
    it = a.iters
    dev = torch.device('cuda', 0)
    from crbhip import sparse, voxel, bnrelu
    from pcdet.datasets.synthetic import kitti_batch, KITTI_RANGE, KITTI_VOXEL
    from pcdet.ops.iou3d_nms import iou3d_nms_utils
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as U
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
    B = 16
    pts_np, off_np, gt_np = kitti_batch(0, B)
    pts, off = torch.from_numpy(pts_np).to(dev), torch.from_numpy(off_np).to(dev)
    n = pts.shape[0]

    # ---- a1/a3 voxel generator + mean
    vox = lambda: voxel.voxelize(pts, off, KITTI_RANGE, KITTI_VOXEL, 16000, 5, want_voxels=False, want_mean=True)
    r = vox()
    M = r['coords'].shape[0]
    row('voxelize + mean (a1,a3): %d pts -> %d voxels' % (n, M), timeit(vox, it), 16 * n + 4 * M * (4 + 4),
        note='6 launches + 1 count read-back')

    # ---- a4 rulebooks and convs per level
    coords, shape = r['coords'], [41, 1600, 1408]
    geo = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1))]
    chans = {1: 16, 2: 32, 3: 64, 4: 64}
    feats = {}
    for lvl in range(1, 5):
        if lvl > 1:
            cprev = coords
            t = timeit(lambda: sparse.spconv_rulebook(cprev, shape, B, *geo[lvl - 2]), max(5, it // 3))
            rbs = sparse.spconv_rulebook(cprev, shape, B, *geo[lvl - 2])
            row('strided rulebook L%d->L%d (out coords + nbr + nbr_t): %d -> %d rows' % (lvl - 1, lvl, rbs.n_in, rbs.n_out),
                t, 16 * rbs.n_in + 4 * 27 * (rbs.n_out + rbs.n_in), note='incl. 1 row-count read-back')
            coords, shape = rbs.out_coords.contiguous(), rbs.out_shape
        cc, ss = coords, shape
        t = timeit(lambda: sparse.subm_rulebook(cc, ss, [3, 3, 3]), max(5, it // 3))
        rb = sparse.subm_rulebook(coords, shape, [3, 3, 3])
        N = rb.n_out
        row('SubM rulebook L%d (hash + nbr): %d rows' % (lvl, N), t, 16 * N + 4 * 27 * N)
        t = timeit(lambda: sparse._compact(rb.nbr, sparse._mask_perm(rb.nbr, 27), 27), max(5, it // 3))
        row('mask sort + tile order + compact table L%d' % lvl, t, 4 * 27 * N + 12 * N + 8 * N + 4 * int((rb.nbr >= 0).sum()))
        C = chans[lvl]
        P = int((rb.nbr >= 0).sum())
        x = torch.randn(N, C, device=dev)
        dy = torch.randn(N, C, device=dev)
        w = torch.randn(27, C, C, device=dev) / 10
        table, pairs = rb.table_for('nbr', C, C), rb.pairs()
        balg = 4.0 * N * C * 2 + 8.0 * P + 4.0 * 27 * C * C
        fl = 2.0 * P * C * C
        row('gather-GEMM fwd/dgrad %dx%d L%d (P=%.2fM)' % (C, C, lvl, P / 1e6),
            timeit(lambda: sparse._conv_forward_raw(x, w, table, N), it), balg, fl)
        row('wgrad %dx%d L%d' % (C, C, lvl), timeit(lambda: sparse._conv_wgrad_raw(x, dy, pairs, 27), it), balg, fl,
            note='plan + MFMA + reduce')
        if C >= 32:
            t3 = timeit(lambda: sparse._conv_forward_raw(x, w, table, N, arithmetic='bf16x3'), it)
            row('OPT-IN bf16x3 gather-GEMM fwd/dgrad %dx%d L%d (W split + 3 bf16 MFMA passes)' % (C, C, lvl), t3, balg,
                note='priced against HBM: 3 bf16 passes need %.1f us of the 2.5 PF MFMA' % (3 * fl / 2.5e15 * 1e6))
        feats[lvl] = (x, N, C)
        # a5 fused BN+ReLU on the sparse rows
        bn = torch.nn.BatchNorm1d(C, eps=1e-3, momentum=0.01).to(dev).train()
        xr = x.clone().requires_grad_(True)

        def bn_f():
            return bnrelu.bn_relu(xr, bn, True)
        z = bn_f()
        row('BN1d+ReLU fwd (stats+apply) L%d: %dx%d' % (lvl, N, C), timeit(bn_f, it), 3 * 4 * N * C)
        g = torch.randn_like(z)
        row('BN1d+ReLU bwd (reduce+apply) L%d' % lvl,
            timeit(lambda: torch.autograd.grad(z, xr, g, retain_graph=True), it), 5 * 4 * N * C)

    # ---- a6 BEV scatter (channels_last), from the conv_out geometry
    rbo = sparse.spconv_rulebook(coords, shape, B, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    oc, osz = rbo.out_coords.contiguous(), rbo.out_shape
    f5 = torch.randn(oc.shape[0], 128, device=dev)
    t = timeit(lambda: sparse.to_bev_channels_last(f5, oc, B, osz), it)
    row('BEV scatter NHWC (memset + rows): %d x 128 -> (16,256,200,176)' % oc.shape[0], t,
        4 * oc.shape[0] * 128 + 4 * B * 256 * 200 * 176, note='dominated by the 577 MB zero-fill')
    # BN on the dense BEV rows (what the BEV backbone uses in training)
    xd = torch.randn(B * 200 * 176, 128, device=dev, requires_grad=True)
    bnd = torch.nn.BatchNorm1d(128, eps=1e-3, momentum=0.01).to(dev).train()
    zd = bnrelu.bn_relu(xd, bnd, True)
    row('BN+ReLU fwd on BEV rows: %d x 128' % xd.shape[0], timeit(lambda: bnrelu.bn_relu(xd, bnd, True), it),
        3 * 4 * xd.numel())
    gd = torch.randn_like(zd)
    row('BN+ReLU bwd on BEV rows', timeit(lambda: torch.autograd.grad(zd, xd, gd, retain_graph=True), it),
        5 * 4 * xd.numel())
    del xd, zd, gd

    # ---- a13/a14 NMS, IoU
    from tests_boxes import detection_boxes
    rng = np.random.default_rng(0)
    for nb, thr, frames in ((9000, 0.8, 16), (4096, 0.1, 16), (1024, 0.7, 16)):
        bx = np.stack([detection_boxes(rng, nb)[0] for _ in range(frames)])
        bt = torch.from_numpy(bx).to(dev)
        cnt = torch.full((frames,), nb, dtype=torch.int32, device=dev)
        t = timeit(lambda: iou3d_nms_utils.nms_batched(bt, cnt, thr, 512), max(5, it // 3))
        row('rotated NMS %d boxes x %d frames (mask + scan), thresh %.2f' % (nb, frames, thr), t,
            frames * (28 * nb + 8 * nb * ((nb + 63) // 64) / 2), note='arithmetic/latency bound: bytes are the mask only')
    a_, b_ = torch.from_numpy(detection_boxes(rng, 512)[0]).to(dev), torch.from_numpy(detection_boxes(rng, 40)[0]).to(dev)
    row('boxes_iou3d 512 x 40', timeit(lambda: iou3d_nms_utils.boxes_iou3d_gpu(a_, b_), it), 28 * 552 + 4 * 512 * 40)

    # ---- a16/a17 PointNet++ ops at PV-RCNN shapes
    xyz = pts[:, :3].contiguous()
    xyzb = xyz.view(B, -1, 3).contiguous()
    row('FPS 16 x 20000 -> 2048', timeit(lambda: U.farthest_point_sample(xyzb, 2048), max(5, it // 3)), None,
        note='2047 serial rounds per frame, one WG per frame')
    kp = torch.gather(xyzb, 1, U.farthest_point_sample(xyzb, 2048).long()[..., None].expand(-1, -1, 3)).reshape(-1, 3).contiguous()
    kc = torch.full((B,), 2048, dtype=torch.int32, device=dev)
    xc = torch.full((B,), xyzb.shape[1], dtype=torch.int32, device=dev)
    row('ball query pair r=0.4/0.8: 32768 keypoints vs 16 x 20000 raw points',
        timeit(lambda: U.ball_query_pair(0.4, 16, 0.8, 16, xyz, xc, kp, kc), it), 12 * (kp.shape[0] + n) + 8 * 16 * kp.shape[0],
        note='VALU-bound: 32768 x 20000 = 655 M distance tests (scan order is part of the contract)')
    grid = (kp.view(B, 2048, 1, 3)[:, :128] + torch.randn(B, 128, 216, 3, device=dev) * 0.7).reshape(-1, 3).contiguous()
    gc = torch.full((B,), 128 * 216, dtype=torch.int32, device=dev)
    row('ball query pair r=0.8/1.6: RoI grid 442368 queries vs 16 x 2048 keypoints',
        timeit(lambda: U.ball_query_pair(0.8, 16, 1.6, 16, kp, kc, grid, gc), it), 12 * (grid.shape[0] + kp.shape[0]) + 8 * 16 * grid.shape[0],
        note='VALU-bound: 442368 x 2048 = 906 M distance tests')
    ball = U.ball_query_pair(0.8, 16, 1.6, 16, kp, kc, grid, gc)[1]
    feat = torch.randn(kp.shape[0], 128, device=dev)
    w1x, w1f, b1 = torch.randn(3, 64, device=dev), torch.randn(128, 64, device=dev) / 11, torch.randn(64, device=dev)
    w2t, b2 = torch.randn(64, 64, device=dev) / 8, torch.randn(64, device=dev)
    out = torch.empty(grid.shape[0], 64, device=dev)
    Mq = grid.shape[0]
    row('fused SA (group + 2-layer MLP + max), RoI grid, one radius',
        timeit(lambda: U.sa_mlp2_max(1.6, 16, kp, kc, grid, gc, feat, w1x, w1f, b1, w2t, b2, out, ball=ball), it),
        4 * (Mq * 16 + Mq * 64 + Mq * 3), 2.0 * Mq * 16 * 64 * 64, note='+ P = F W1f^T (32768 x 128 x 64 GEMM)')
    xg = U.query_and_group_rows(1.6, 16, kp, kc, grid, gc, feat, ball=ball)[0]
    row('grouping (row-major, training) RoI grid: (%d, 131)' % xg.shape[0],
        timeit(lambda: U.query_and_group_rows(1.6, 16, kp, kc, grid, gc, feat, ball=ball), it), 4 * xg.numel() + 4 * Mq * 16)
    del xg
    # ---- a15 points in boxes
    gtb = torch.from_numpy(gt_np).to(dev)[..., :7].contiguous()
    row('points_in_boxes 16 x 20000 pts x %d boxes' % gtb.shape[1],
        timeit(lambda: roiaware_pool3d_utils.points_in_boxes_gpu(xyzb, gtb), it), 12 * n + 28 * gtb.shape[0] * gtb.shape[1] + 4 * n)

    print('| kernel (BASELINE configs[1] size) | median µs | algorithmic bytes | achieved | f32 flops | bound, fraction of roof | note |')
    print('|---|---|---|---|---|---|---|')
    print('\n'.join(ROWS))


if __name__ == '__main__':
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import boxes_synth as tests_boxes
    sys.modules['tests_boxes'] = tests_boxes
    main()
