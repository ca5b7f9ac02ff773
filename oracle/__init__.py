"""ORACLE — test infrastructure only.

CPU restatements (plain C via ctypes + numpy) of the reference algorithms on the hot path. Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package; the product
path (crb-active-3ddet_amd/) never does. Each function cites the reference file:line it follows in the
C source next to it. See DESIGN.md §Oracle for what pins each piece.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith('_oracle.c')]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'liboracle.so'])


def _lib():
    global _L
    try:
        return _L
    except NameError:
        pass
    build()
    _L = ctypes.CDLL(_SO)
    return _L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _arr3(v, dt=np.int32):
    return np.ascontiguousarray(np.asarray(v).reshape(3), dtype=dt)


# ------------------------------------------------------------------ voxelizer
def voxelize_frame(points, range_min_xyz, voxel_size_xyz, grid_xyz, max_voxels, max_points):
    """-> voxels (M,max_points,C), coords (M,3)[z,y,x], num_points (M)"""
    L = _lib()
    pts = _f32(points)
    n, C = pts.shape
    voxels = np.empty((max_voxels, max_points, C), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    npts = np.zeros((max_voxels,), np.int32)
    L.oracle_voxelize_frame.restype = ctypes.c_int
    m = L.oracle_voxelize_frame(_p(pts), ctypes.c_int(n), ctypes.c_int(C), _p(_arr3(range_min_xyz, np.float32)),
                                _p(_arr3(voxel_size_xyz, np.float32)), _p(_arr3(grid_xyz)), ctypes.c_int(max_voxels),
                                ctypes.c_int(max_points), _p(voxels), _p(coords), _p(npts))
    assert m >= 0
    return voxels[:m].copy(), coords[:m].copy(), npts[:m].copy()


def voxelize_batch(points, frame_offsets, range_min_xyz, voxel_size_xyz, grid_xyz, max_voxels, max_points):
    """collate like pcdet/datasets/dataset.py:160-229: coords (M,4) [b,z,y,x]"""
    vs, cs, ns, counts = [], [], [], []
    for b in range(len(frame_offsets) - 1):
        v, c, k = voxelize_frame(points[frame_offsets[b]:frame_offsets[b + 1]], range_min_xyz, voxel_size_xyz, grid_xyz,
                                 max_voxels, max_points)
        vs.append(v)
        cs.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], axis=1))
        ns.append(k)
        counts.append(len(c))
    return np.concatenate(vs), np.concatenate(cs), np.concatenate(ns), np.asarray(counts, np.int32)


def mean_vfe(voxels, num_points):
    L = _lib()
    v = _f32(voxels)
    M, T, C = v.shape
    out = np.empty((M, C), np.float32)
    L.oracle_mean_vfe(_p(v), _p(_i32(num_points)), ctypes.c_int(M), ctypes.c_int(T), ctypes.c_int(C), _p(out))
    return out


# ------------------------------------------------------------------ sparse conv
def subm_nbr(coords, shape_dhw, ksize):
    L = _lib()
    c = _i32(coords)
    ks = _arr3(ksize)
    K = int(ks.prod())
    nbr = np.empty((len(c), K), np.int32)
    L.oracle_subm_nbr(_p(c), ctypes.c_int(len(c)), _p(_arr3(shape_dhw)), _p(ks), _p(nbr))
    return nbr


def conv_out_shape(shape_dhw, ksize, stride, padding):
    return [(int(s) + 2 * int(p) - int(k)) // int(st) + 1 for s, k, st, p in zip(shape_dhw, ksize, stride, padding)]


def spconv_out(coords, shape_dhw, ksize, stride, padding):
    """-> out_coords (n_out,4) ascending (b,z,y,x), out_shape"""
    L = _lib()
    c = _i32(coords)
    ks, st, pd = _arr3(ksize), _arr3(stride), _arr3(padding)
    oshape = _arr3(conv_out_shape(shape_dhw, ks, st, pd))
    cap = max(1, len(c) * int(ks.prod()))
    out = np.empty((cap, 4), np.int32)
    L.oracle_spconv_out.restype = ctypes.c_int
    n_out = L.oracle_spconv_out(_p(c), ctypes.c_int(len(c)), _p(ks), _p(st), _p(pd), _p(oshape), _p(out),
                                ctypes.c_int(cap))
    return out[:n_out].copy(), [int(v) for v in oshape]


def spconv_nbr(coords, shape_dhw, out_coords, ksize, stride, padding):
    L = _lib()
    c, oc = _i32(coords), _i32(out_coords)
    ks, st, pd = _arr3(ksize), _arr3(stride), _arr3(padding)
    nbr = np.empty((len(oc), int(ks.prod())), np.int32)
    L.oracle_spconv_nbr(_p(c), ctypes.c_int(len(c)), _p(_arr3(shape_dhw)), _p(oc), ctypes.c_int(len(oc)), _p(ks), _p(st),
                        _p(pd), _p(nbr))
    return nbr


def conv_fwd(X, W, nbr):
    """X (n_in,cin), W (K,cin,cout), nbr (n_out,K) -> Y (n_out,cout)"""
    L = _lib()
    X, W, nbr = _f32(X), _f32(W), _i32(nbr)
    K, cin, cout = W.shape
    Y = np.empty((len(nbr), cout), np.float32)
    L.oracle_conv_fwd(_p(X), _p(W), _p(nbr), _p(Y), ctypes.c_int(len(nbr)), ctypes.c_int(K), ctypes.c_int(cin),
                      ctypes.c_int(cout))
    return Y


def conv_dgrad(dY, W, nbr, n_in):
    L = _lib()
    dY, W, nbr = _f32(dY), _f32(W), _i32(nbr)
    K, cin, cout = W.shape
    dX = np.empty((n_in, cin), np.float32)
    L.oracle_conv_dgrad(_p(dY), _p(W), _p(nbr), _p(dX), ctypes.c_int(n_in), ctypes.c_int(len(nbr)), ctypes.c_int(K),
                        ctypes.c_int(cin), ctypes.c_int(cout))
    return dX


def conv_wgrad(X, dY, nbr, K):
    L = _lib()
    X, dY, nbr = _f32(X), _f32(dY), _i32(nbr)
    cin, cout = X.shape[1], dY.shape[1]
    dW = np.empty((K, cin, cout), np.float32)
    L.oracle_conv_wgrad(_p(X), _p(dY), _p(nbr), _p(dW), ctypes.c_int(len(nbr)), ctypes.c_int(K), ctypes.c_int(cin),
                        ctypes.c_int(cout))
    return dW


def dense(feat, coords, B, shape_dhw):
    L = _lib()
    f, c = _f32(feat), _i32(coords)
    C = f.shape[1]
    D, H, W = [int(v) for v in shape_dhw]
    out = np.empty((B, C, D, H, W), np.float32)
    L.oracle_dense(_p(f), _p(c), _p(out), ctypes.c_int(len(c)), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(D),
                   ctypes.c_int(H), ctypes.c_int(W))
    return out


# ------------------------------------------------------------------ rotated IoU / NMS
def boxes_pairwise(boxes_a, boxes_b, mode):
    """mode 0: BEV overlap area, 1: BEV IoU, 2: 3-D IoU"""
    L = _lib()
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((len(a), len(b)), np.float32)
    L.oracle_boxes_pairwise(_p(a), ctypes.c_int(len(a)), _p(b), ctypes.c_int(len(b)), _p(out), ctypes.c_int(mode))
    return out


def nms(boxes_sorted, thresh, rotated=True):
    """greedy NMS over boxes already in descending score order -> kept indices"""
    L = _lib()
    b = _f32(boxes_sorted)
    keep = np.zeros((max(len(b), 1),), np.int32)
    L.oracle_nms.restype = ctypes.c_int
    n = L.oracle_nms(_p(b), ctypes.c_int(len(b)), ctypes.c_float(thresh), ctypes.c_int(1 if rotated else 0), _p(keep))
    return keep[:n].copy()


# ------------------------------------------------------------------ oracle/_ref : the reference's own compiled code
_REF_IOU = os.path.join(_HERE, '_ref', 'libiou3d_ref.so')


def have_ref_iou3d():
    return os.path.exists(_REF_IOU)


def ref_boxes_iou_bev(boxes_a, boxes_b):
    """the reference's boxes_iou_bev_cpu (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252) compiled from its own source
    by oracle/build_ref.sh"""
    import torch  # noqa: F401  (libtorch must be resident before the shim loads)
    global _RL
    try:
        L = _RL
    except NameError:
        L = _RL = ctypes.CDLL(_REF_IOU)
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((len(a), len(b)), np.float32)
    L.ref_boxes_iou_bev_cpu(_p(a), ctypes.c_int(len(a)), _p(b), ctypes.c_int(len(b)), _p(out))
    return out


def _ref_roiaware_path():
    import sysconfig
    return os.path.join(_HERE, '_ref', 'roiaware_pool3d_ref' + sysconfig.get_config_var('EXT_SUFFIX'))


def have_ref_roiaware():
    return os.path.exists(_ref_roiaware_path())


def ref_points_in_boxes_cpu(boxes, points):
    """the reference's points_in_boxes_cpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:144-167) called through
    its own pybind module, compiled from its own source by oracle/build_ref.sh. The module also declares CUDA launchers
    that are never called: it is imported with RTLD_LAZY so they stay unresolved. -> (N,P) int32 membership"""
    import importlib.util
    import sys
    import torch
    global _RR
    try:
        mod = _RR
    except NameError:
        flags = sys.getdlopenflags()
        sys.setdlopenflags(os.RTLD_LAZY)
        try:
            spec = importlib.util.spec_from_file_location('roiaware_pool3d_ref', _ref_roiaware_path())
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            sys.setdlopenflags(flags)
        _RR = mod
    b, p = torch.from_numpy(_f32(boxes)), torch.from_numpy(_f32(points))
    out = torch.zeros((b.shape[0], p.shape[0]), dtype=torch.int32)
    mod.points_in_boxes_cpu(b, p, out)
    return out.numpy()


def points_in_boxes_cpu(boxes, points):
    """restatement of the CPU twin (1e-2 margin) -> (N,P) int32 membership"""
    L = _lib()
    b, p = _f32(boxes), _f32(points)
    out = np.empty((len(b), len(p)), np.int32)
    L.oracle_points_in_boxes_cpu(_ci(len(b)), _ci(len(p)), _p(b), _p(p), _p(out))
    return out


# ------------------------------------------------------------------ PointNet++ stack ops, points-in-boxes, RoI-aware pool
def _ci(v):
    return ctypes.c_int(int(v))


def ball_query(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt):
    """-> idx (M,nsample) as written by the kernel (empty ball: [-1,0,0,..])"""
    L = _lib()
    xyz, new_xyz, xyz_cnt, new_cnt = _f32(xyz), _f32(new_xyz), _i32(xyz_cnt), _i32(new_cnt)
    idx = np.zeros((len(new_xyz), nsample), np.int32)
    L.oracle_ball_query(_ci(len(xyz_cnt)), _ci(len(new_xyz)), ctypes.c_float(radius), _ci(nsample), _p(new_xyz),
                        _p(new_cnt), _p(xyz), _p(xyz_cnt), _p(idx))
    return idx


def group_points(feat, feat_cnt, idx, idx_cnt):
    L = _lib()
    feat, idx, feat_cnt, idx_cnt = _f32(feat), _i32(idx), _i32(feat_cnt), _i32(idx_cnt)
    M, ns = idx.shape
    C = feat.shape[1]
    out = np.empty((M, C, ns), np.float32)
    L.oracle_group_points(_ci(len(idx_cnt)), _ci(M), _ci(C), _ci(ns), _p(feat), _p(feat_cnt), _p(idx), _p(idx_cnt), _p(out))
    return out


def group_points_grad(grad_out, idx, idx_cnt, feat_cnt, N):
    L = _lib()
    g, idx, feat_cnt, idx_cnt = _f32(grad_out), _i32(idx), _i32(feat_cnt), _i32(idx_cnt)
    M, C, ns = g.shape
    out = np.empty((N, C), np.float32)
    L.oracle_group_points_grad(_ci(len(idx_cnt)), _ci(M), _ci(C), _ci(N), _ci(ns), _p(g), _p(idx), _p(idx_cnt),
                               _p(feat_cnt), _p(out))
    return out


def fps(xyz, m):
    """xyz (B,n,3) -> (B,m) int32"""
    L = _lib()
    xyz = _f32(xyz)
    B, n, _ = xyz.shape
    out = np.zeros((B, m), np.int32)
    L.oracle_fps(_ci(B), _ci(n), _ci(m), _p(xyz), _p(out))
    return out


def three_nn(unknown, unknown_cnt, known, known_cnt):
    L = _lib()
    u, k, uc, kc = _f32(unknown), _f32(known), _i32(unknown_cnt), _i32(known_cnt)
    d2 = np.empty((len(u), 3), np.float32)
    idx = np.empty((len(u), 3), np.int32)
    L.oracle_three_nn(_ci(len(uc)), _ci(len(u)), _p(u), _p(uc), _p(k), _p(kc), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(feat, idx, w):
    L = _lib()
    f, i, w = _f32(feat), _i32(idx), _f32(w)
    out = np.empty((len(i), f.shape[1]), np.float32)
    L.oracle_three_interpolate(_ci(len(i)), _ci(f.shape[1]), _p(f), _p(i), _p(w), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, w, M):
    L = _lib()
    g, i, w = _f32(grad_out), _i32(idx), _f32(w)
    out = np.empty((M, g.shape[1]), np.float32)
    L.oracle_three_interpolate_grad(_ci(len(i)), _ci(g.shape[1]), _ci(M), _p(g), _p(i), _p(w), _p(out))
    return out


def points_in_boxes(points, boxes):
    """points (B,M,3), boxes (B,T,7) -> (B,M) int32"""
    L = _lib()
    p, b = _f32(points), _f32(boxes)
    out = np.empty(p.shape[:2], np.int32)
    L.oracle_points_in_boxes(_ci(p.shape[0]), _ci(b.shape[1]), _ci(p.shape[1]), _p(b), _p(p), _p(out))
    return out


def roiaware_pool(rois, pts, feat, out_size, max_pts, method):
    L = _lib()
    r, p, f = _f32(rois), _f32(pts), _f32(feat)
    ox, oy, oz = out_size
    N, C = len(r), f.shape[1]
    argmax = np.empty((N, ox, oy, oz, C), np.int32)
    pts_idx = np.empty((N, ox, oy, oz, max_pts), np.int32)
    pooled = np.empty((N, ox, oy, oz, C), np.float32)
    L.oracle_roiaware_pool(_ci(N), _ci(len(p)), _ci(C), _ci(max_pts), _ci(ox), _ci(oy), _ci(oz), _p(r), _p(p), _p(f),
                           _p(argmax), _p(pts_idx), _p(pooled), _ci(method))
    return pooled, argmax, pts_idx
