"""pcdet.ops.pointnet2.pointnet2_stack.pointnet2_utils (reference: pointnet2_utils.py:8-299) over csrc/pointnet2_stack.hip.
Same callables: ball_query, grouping_operation, QueryAndGroup, farthest_point_sample, stack_farthest_point_sample,
three_nn, three_interpolate."""
import ctypes

import math
import torch
import torch.nn as nn
from torch.autograd import Function

from crbhip import lib, check, ptr, cur_stream, require_cuda
from ....utils.linear_rows import tall_t_matmul as _tall_t_matmul


def _i32(t):
    return t.to(torch.int32).contiguous()


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        """xyz (N,3), xyz_batch_cnt (B), new_xyz (M,3), new_xyz_batch_cnt (B) -> idx (M,nsample) int32, empty (M) bool"""
        require_cuda(xyz, new_xyz)
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        idx = torch.empty((M, nsample), dtype=torch.int32, device=xyz.device)
        check(lib.crb_ball_query_stack(B, M, float(radius), int(nsample), ptr(new_xyz), ptr(_i32(new_xyz_batch_cnt)),
                                       ptr(xyz), ptr(_i32(xyz_batch_cnt)), ptr(idx), cur_stream(xyz.device)),
              'crb_ball_query_stack')
        empty_ball_mask = (idx[:, 0] == -1)
        idx = torch.where(empty_ball_mask[:, None], torch.zeros_like(idx), idx)
        ctx.mark_non_differentiable(idx, empty_ball_mask)
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


# CRB_BALL_QUERY_GRID=1 (opt-in): the ball queries of the set-abstraction layers on a per-call cell grid (crb_ball_query2_grid_stack)
# instead of a scan of the whole frame. Index-exact (tests), but measured SLOWER than the 8-queries-per-wave scans at 20,000 points per
# frame on every source of the PV-RCNN set abstraction (tools/time_ball_query.py: 198-371 us against 51-293 us per call at 16 frames,
# 661-1,222 against 143-891 at 64): one query per wave pays two dependent random loads per candidate and a selection pass, the scan
# streams coalesced points for eight queries at once and ends early. Frames with fewer points than the threshold are always scanned
BALL_QUERY_GRID = __import__('os').environ.get('CRB_BALL_QUERY_GRID', '0') == '1'
BALL_QUERY_GRID_MIN_POINTS = int(__import__('os').environ.get('CRB_BALL_QUERY_GRID_MIN_POINTS', '2048'))
_BQ_WS = {}


@torch.no_grad()
def ball_query_pair(radius_a, nsample_a, radius_b, nsample_b, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, group=None):
    """both radii of a StackSAModuleMSG in one scan -> ((idx_a, empty_a), (idx_b, empty_b)), the same values as two
    ball_query calls (idx int32 with empty balls zeroed, empty as uint8). `group`: the queries come in spatially compact groups
    of that many consecutive rows, none straddling a frame (RoI grid points): crb_ball_query2_grouped_stack prefilters the
    frame's points per group — same result."""
    require_cuda(xyz, new_xyz)
    xyz, new_xyz = xyz.contiguous(), new_xyz.contiguous()
    B, M, dev = xyz_batch_cnt.shape[0], new_xyz.shape[0], xyz.device
    ia = torch.empty((M, nsample_a), dtype=torch.int32, device=dev)
    ib = torch.empty((M, nsample_b), dtype=torch.int32, device=dev)
    ea = torch.empty((M,), dtype=torch.uint8, device=dev)
    eb = torch.empty((M,), dtype=torch.uint8, device=dev)
    if group and 0 < group <= 1024 and M % group == 0:
        check(lib.crb_ball_query2_grouped_stack(B, M, int(group), float(radius_a), int(nsample_a), float(radius_b),
                                                int(nsample_b), ptr(new_xyz), ptr(_i32(new_xyz_batch_cnt)), ptr(xyz),
                                                ptr(_i32(xyz_batch_cnt)), ptr(ia), ptr(ib), ptr(ea), ptr(eb),
                                                cur_stream(dev)), 'crb_ball_query2_grouped_stack')
        return (ia, ea), (ib, eb)
    N = xyz.shape[0]
    if BALL_QUERY_GRID and radius_a <= radius_b and N >= B * BALL_QUERY_GRID_MIN_POINTS and max(nsample_a, nsample_b) <= 64:
        # the call's points counting-sorted into cells of the larger radius, 27 cells per query: the same lists, index for index
        nbytes = int(lib.crb_ball_query2_grid_workspace_bytes(N))
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _BQ_WS.get(key)                      # one per (device, stream): every call on a stream is ordered behind the last
        if ws is None or ws.numel() * 4 < nbytes:
            ws = _BQ_WS[key] = torch.empty((nbytes // 4 + 64,), dtype=torch.int32, device=dev)
        check(lib.crb_ball_query2_grid_stack(B, M, float(radius_a), int(nsample_a), float(radius_b), int(nsample_b), ptr(new_xyz),
                                             ptr(_i32(new_xyz_batch_cnt)), ptr(xyz), ptr(_i32(xyz_batch_cnt)), N, ptr(ia), ptr(ib),
                                             ptr(ea), ptr(eb), ptr(ws), ws.numel() * 4, cur_stream(dev)), 'crb_ball_query2_grid_stack')
        return (ia, ea), (ib, eb)
    check(lib.crb_ball_query2_stack(B, M, float(radius_a), int(nsample_a), float(radius_b), int(nsample_b), ptr(new_xyz),
                                    ptr(_i32(new_xyz_batch_cnt)), ptr(xyz), ptr(_i32(xyz_batch_cnt)), ptr(ia), ptr(ib),
                                    ptr(ea), ptr(eb), cur_stream(dev)), 'crb_ball_query2_stack')
    return (ia, ea), (ib, eb)


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        """features (N,C), idx (M,nsample) -> (M,C,nsample)"""
        require_cuda(features, idx)
        assert features.is_contiguous() and idx.is_contiguous()
        M, nsample = idx.shape
        N, C = features.shape
        B = idx_batch_cnt.shape[0]
        fcnt, icnt = _i32(features_batch_cnt), _i32(idx_batch_cnt)
        out = torch.empty((M, C, nsample), dtype=torch.float32, device=features.device)
        check(lib.crb_group_points_stack(B, M, C, nsample, ptr(features), ptr(fcnt), ptr(idx), ptr(icnt), ptr(out),
                                         cur_stream(features.device)), 'crb_group_points_stack')
        ctx.for_backwards = (B, N, idx, fcnt, icnt)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, fcnt, icnt = ctx.for_backwards
        M, C, nsample = grad_out.shape
        g = grad_out.contiguous().float()
        grad_features = torch.zeros((N, C), dtype=torch.float32, device=g.device)
        check(lib.crb_group_points_grad_stack(B, M, C, nsample, ptr(g), ptr(idx), ptr(icnt), ptr(fcnt),
                                              ptr(grad_features), cur_stream(g.device)), 'crb_group_points_grad_stack')
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """-> new_features (M, 3+C, nsample), idx (M, nsample)   (pointnet2_utils.py:107-155)"""
        idx, empty = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt) - new_xyz.unsqueeze(-1)
        keep = (~empty).view(-1, 1, 1).to(grouped_xyz.dtype)
        grouped_xyz = grouped_xyz * keep
        if features is not None:
            grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt) * keep
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = grouped_xyz
        return new_features, idx


class FusedQueryGroup(Function):
    """ball-query result -> (1, 3+C, M, nsample) tensor ready for the shared MLP (one HIP launch)"""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty):
        require_cuda(xyz, new_xyz, features, idx)
        M, ns = idx.shape
        N, C = features.shape
        B = xyz_batch_cnt.shape[0]
        xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
        em = empty.to(torch.uint8).contiguous()
        out = torch.empty((1, 3 + C, M, ns), dtype=torch.float32, device=xyz.device)
        check(lib.crb_query_group_stack(B, M, C, ns, ptr(xyz.contiguous()), ptr(xc), ptr(features.contiguous().float()),
                                        ptr(new_xyz.contiguous()), ptr(nc), ptr(idx.contiguous()), ptr(em), ptr(out),
                                        cur_stream(xyz.device)), 'crb_query_group_stack')
        ctx.meta = (B, M, C, ns, N, xc, nc, idx, em)
        return out

    @staticmethod
    def backward(ctx, g):
        B, M, C, ns, N, xc, nc, idx, em = ctx.meta
        g = g.contiguous().float()
        gf = torch.zeros((N, C), dtype=torch.float32, device=g.device)
        check(lib.crb_query_group_grad_stack(B, M, C, ns, ptr(xc), ptr(nc), ptr(idx), ptr(em), ptr(g), ptr(gf),
                                             cur_stream(g.device)), 'crb_query_group_grad_stack')
        return None, None, None, None, gf, None, None


def query_and_group_fused(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
    """-> (1, 3+C, M, nsample), idx (M, nsample)"""
    idx, empty = ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    return FusedQueryGroup.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty), idx


class FusedQueryGroupRows(Function):
    """ball-query result -> (M*nsample, 3+C) row-major matrix [xyz[nbr]-new_xyz ; features[nbr]] (one HIP launch)"""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty):
        require_cuda(xyz, new_xyz, features, idx)
        M, ns = idx.shape
        N, C = features.shape
        B = xyz_batch_cnt.shape[0]
        xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
        em = empty.to(torch.uint8).contiguous()
        out = torch.empty((M * ns, 3 + C), dtype=torch.float32, device=xyz.device)
        check(lib.crb_query_group_rows_stack(B, M, C, ns, ptr(xyz.contiguous()), ptr(xc),
                                             ptr(features.contiguous().float()), ptr(new_xyz.contiguous()), ptr(nc),
                                             ptr(idx.contiguous()), ptr(em), ptr(out), cur_stream(xyz.device)),
              'crb_query_group_rows_stack')
        ctx.meta = (B, M, C, ns, N, xc, nc, idx, em)
        return out

    @staticmethod
    def backward(ctx, g):
        B, M, C, ns, N, xc, nc, idx, em = ctx.meta
        g = g.contiguous().float()
        gf = torch.zeros((N, C), dtype=torch.float32, device=g.device)
        check(lib.crb_query_group_rows_grad_stack(B, M, C, ns, ptr(xc), ptr(nc), ptr(idx), ptr(em), ptr(g), ptr(gf),
                                                  cur_stream(g.device)), 'crb_query_group_rows_grad_stack')
        return None, None, None, None, gf, None, None


def query_and_group_rows(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, ball=None):
    """-> (M*nsample, 3+C), idx (M, nsample); `ball` = (idx, empty) from ball_query_pair skips the query"""
    idx, empty = ball if ball is not None else ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    return FusedQueryGroupRows.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty), idx


class GroupedFirstLayerRows(Function):
    """y (M*nsample, H) = [xyz[nbr]-new_xyz ; features[nbr]] @ weight^T without forming the grouped matrix: the feature part of
    the product is taken per SOURCE point (P = features @ W1f^T, N rows) and gathered (crb_group_affine_rows_stack)."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty, weight):
        require_cuda(xyz, new_xyz, features, idx, weight)
        M, ns = idx.shape
        H = weight.shape[0]
        B = xyz_batch_cnt.shape[0]
        xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
        em = empty.to(torch.uint8).contiguous()
        idx = idx.contiguous()
        feats = features.contiguous().float()
        w1x = weight[:, :3].t().contiguous()                  # (3, H)
        w1f = weight[:, 3:].contiguous()                      # (H, C)
        P = feats @ w1f.t()
        xyz_c, new_c = xyz.contiguous(), new_xyz.contiguous()
        out = torch.empty((M * ns, H), dtype=torch.float32, device=xyz.device)
        rel = torch.empty((M * ns, 3), dtype=torch.float32, device=xyz.device)
        check(lib.crb_group_affine_rows_stack(B, M, H, ns, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc), ptr(idx),
                                              ptr(em), ptr(w1x), ptr(out), ptr(rel), cur_stream(xyz.device)),
              'crb_group_affine_rows_stack')
        ctx.meta = (B, M, H, ns, xc, nc, idx, em)
        ctx.save_for_backward(feats, w1f, rel)
        return out

    @staticmethod
    def backward(ctx, g):
        B, M, H, ns, xc, nc, idx, em = ctx.meta
        feats, w1f, rel = ctx.saved_tensors
        g = g.contiguous().float()
        gP = torch.zeros((feats.shape[0], H), dtype=torch.float32, device=g.device)
        part = torch.empty((int(lib.crb_group_affine_rows_grad_blocks(M, ns)), 3, H), dtype=torch.float32, device=g.device)
        check(lib.crb_group_affine_rows_grad_stack(B, M, H, ns, ptr(xc), ptr(nc), ptr(idx), ptr(em), ptr(rel), ptr(g),
                                                   ptr(gP), ptr(part), cur_stream(g.device)),
              'crb_group_affine_rows_grad_stack')
        gf = gP @ w1f if ctx.needs_input_grad[4] else None
        gw = None
        if ctx.needs_input_grad[7]:
            gw = torch.cat([part.sum(0).t(), _tall_t_matmul(gP, feats)], dim=1)           # (H, 3+C)
        return None, None, None, None, gf, None, None, gw


FIRST_LAYER_SLAB_STATS = __import__('os').environ.get('CRB_SLAB_STATS', '1') == '1'    # GroupedFirstLayerBNReLU: BatchNorm statistics from slab sums written by the producer kernel


class GroupedFirstLayerBNReLU(Function):
    """relu(batchnorm(GroupedFirstLayerRows(...))) in training mode as ONE autograd node: forward = the two launches' worth of
    kernels the separate ops run; backward reduces dgamma / dbeta (crb_bn_relu_backward with dx = NULL) and hands them to
    crb_group_affine_rows_grad_bn_stack, which applies the BatchNorm backward while it loads the gradient slab — the
    (M*nsample, H) gradient of the layer's output (1.8 GB at the RoI-grid shape) is neither written nor read."""

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty, weight, gamma, beta, eps,
                running_mean, running_var, momentum, nbt):
        from crbhip import bnrelu
        require_cuda(xyz, new_xyz, features, idx, weight, gamma, beta)
        M, ns = idx.shape
        H = weight.shape[0]
        B = xyz_batch_cnt.shape[0]
        dev = xyz.device
        xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
        em = empty.to(torch.uint8).contiguous()
        idx = idx.contiguous()
        feats = features.contiguous().float()
        w1x = weight[:, :3].t().contiguous()
        w1f = weight[:, 3:].contiguous()
        P = feats @ w1f.t()
        y = torch.empty((M * ns, H), dtype=torch.float32, device=dev)
        rel = torch.empty((M * ns, 3), dtype=torch.float32, device=dev)
        n = M * ns
        z = torch.empty_like(y)
        mean = torch.empty((H,), dtype=torch.float32, device=dev)
        var, invstd = torch.empty_like(mean), torch.empty_like(mean)
        wsb = lib.crb_bn_workspace_bytes(n, H)
        ws, tk = bnrelu._scratch(dev, wsb)
        g, b = gamma.contiguous().float(), beta.contiguous().float()
        if FIRST_LAYER_SLAB_STATS and H in (16, 32, 64, 128):
            # the producer writes the column sums of every 64-row slab: the BatchNorm's statistics pass reads those (2 H floats
            # per 64 rows) instead of the (M*ns, H) rows
            nslab = int(lib.crb_group_affine_rows_grad_blocks(M, ns))
            stat = torch.empty((nslab, 2, H), dtype=torch.float32, device=dev)
            check(lib.crb_group_affine_rows_stats_stack(B, M, H, ns, ptr(xyz.contiguous()), ptr(xc), ptr(P),
                                                        ptr(new_xyz.contiguous()), ptr(nc), ptr(idx), ptr(em), ptr(w1x), ptr(y),
                                                        ptr(rel), ptr(stat), cur_stream(dev)), 'crb_group_affine_rows_stats_stack')
            check(lib.crb_bn_relu_forward_partials(ptr(y), n, H, ptr(stat), nslab, ptr(g), ptr(b), float(eps), 1, ptr(z), 0,
                                                   ptr(mean), ptr(var), ptr(invstd), ptr(running_mean), ptr(running_var),
                                                   ptr(nbt), float(momentum), ptr(ws), wsb, ptr(tk), cur_stream(dev)),
                  'crb_bn_relu_forward_partials')
        else:
            check(lib.crb_group_affine_rows_stack(B, M, H, ns, ptr(xyz.contiguous()), ptr(xc), ptr(P), ptr(new_xyz.contiguous()),
                                                  ptr(nc), ptr(idx), ptr(em), ptr(w1x), ptr(y), ptr(rel), cur_stream(dev)),
                  'crb_group_affine_rows_stack')
            check(lib.crb_bn_relu_forward(ptr(y), n, H, ptr(g), ptr(b), float(eps), 1, ptr(z), 0, ptr(mean), ptr(var),
                                          ptr(invstd), ptr(running_mean), ptr(running_var), ptr(nbt), float(momentum), ptr(ws),
                                          wsb, ptr(tk), cur_stream(dev)), 'crb_bn_relu_forward')
        bnrelu._touch(running_mean, running_var, nbt)
        ctx.meta = (B, M, H, ns, xc, nc, idx, em)
        ctx.save_for_backward(feats, w1f, rel, y, mean, invstd, g, b)
        return z

    @staticmethod
    def backward(ctx, gz):
        from crbhip import bnrelu
        B, M, H, ns, xc, nc, idx, em = ctx.meta
        feats, w1f, rel, y, mean, invstd, g, b = ctx.saved_tensors
        dev = gz.device
        gz = gz.contiguous().float()
        n = M * ns
        dgamma = torch.empty((H,), dtype=torch.float32, device=dev)
        dbeta = torch.empty_like(dgamma)
        wsb = lib.crb_bn_workspace_bytes(n, H)
        ws, tk = bnrelu._scratch(dev, wsb)
        check(lib.crb_bn_relu_backward(ptr(y), ptr(gz), 0, n, H, ptr(mean), ptr(invstd), ptr(g), ptr(b), 1, None, ptr(dgamma),
                                       ptr(dbeta), ptr(ws), wsb, ptr(tk), cur_stream(dev)), 'crb_bn_relu_backward')
        gP = torch.zeros((feats.shape[0], H), dtype=torch.float32, device=dev)
        part = torch.empty((int(lib.crb_group_affine_rows_grad_blocks(M, ns)), 3, H), dtype=torch.float32, device=dev)
        check(lib.crb_group_affine_rows_grad_bn_stack(B, M, H, ns, ptr(xc), ptr(nc), ptr(idx), ptr(em), ptr(rel), ptr(gz), ptr(y),
                                                      ptr(mean), ptr(invstd), ptr(g), ptr(b), ptr(dbeta), ptr(dgamma), ptr(gP),
                                                      ptr(part), cur_stream(dev)), 'crb_group_affine_rows_grad_bn_stack')
        gf = gP @ w1f if ctx.needs_input_grad[4] else None
        gw = torch.cat([part.sum(0).t(), _tall_t_matmul(gP, feats)], dim=1) if ctx.needs_input_grad[7] else None
        return None, None, None, None, gf, None, None, gw, dgamma, dbeta, None, None, None, None, None


def grouped_first_layer_bn_relu(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, weight, bn, ball=None):
    """training-mode relu(bn(first bias-free 1x1 conv of a StackSAModuleMSG scale on the ball-query groups)) -> (M*nsample, H);
    bn: nn.BatchNorm2d / 1d in training mode with a momentum (running statistics and the batch counter are updated)"""
    from crbhip import bnrelu
    idx, empty = ball if ball is not None else ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    return GroupedFirstLayerBNReLU.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty, weight, bn.weight,
                                         bn.bias, bn.eps, bn.running_mean, bn.running_var, float(bn.momentum), bnrelu._counter(bn))


def grouped_first_layer_rows(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, weight, ball=None):
    """first bias-free 1x1 conv of a StackSAModuleMSG scale applied to the ball-query groups -> (M*nsample, H)"""
    idx, empty = ball if ball is not None else ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    return GroupedFirstLayerRows.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, idx, empty, weight)


# SAMlp2TrainConcat backward: scatter of the first layer's gradient in source-row order (crb_pair_sort_by_source) for layers of at least
# this many (query, sample) pairs - the RoI-grid scales: 7 M pairs onto 32 k keypoints, where the real step's scatter takes 0.7 / 1.9 ms
# in pair order (hot keypoints shared by thousands of grid points). Below it the device sort (a merge sort of ~19 launches for
# n < 2 M) costs more than the kernel gains (139 -> 69 us at the voxel levels). CRB_SA_SORTED_SCATTER=0 = always pair order (A/B)
SORTED_SCATTER = __import__('os').environ.get('CRB_SA_SORTED_SCATTER', '1') == '1'
SORTED_SCATTER_MIN_PAIRS = 2 * 1024 * 1024


def sa_mlp2_train_supported(h1, h2, nsample):
    return bool(lib.crb_sa_mlp2_train_supported(int(h1), int(h2), int(nsample)))


class SAMlp2TrainConcat(Function):
    """Training-mode StackSAModuleMSG body for two-layer shared MLPs, all scales of the module, as ONE autograd node writing
    one (M, sum h2) matrix (pointnet2_modules.py:90-112: grouping -> Conv-BN-ReLU -> Conv-BN-ReLU -> max over nsample -> cat)
    with no (M*nsample, h) activation kept or written in the forward: csrc/sa_mlp_train.hip (statistics passes recompute the
    layers from the ball-query indices). The backward recomputes once more and materialises only the masked gradient of the
    first layer's activation, one scale at a time.
    args: xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, then per scale
      idx, empty, W1 (h1, 3+C), gamma1, beta1, eps1, running_mean1, running_var1, momentum1, counter1,
      W2 (h2, h1), gamma2, beta2, eps2, running_mean2, running_var2, momentum2, counter2"""
    PER = 18

    @staticmethod
    def forward(ctx, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, *args):
        from crbhip import bnrelu
        PER = SAMlp2TrainConcat.PER
        k = len(args) // PER
        require_cuda(xyz, new_xyz, features)
        dev = xyz.device
        B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
        xyz_c, new_c = xyz.contiguous().float(), new_xyz.contiguous().float()
        feats = features.contiguous().float()
        widths = [args[PER * i + 10].shape[0] for i in range(k)]
        total = sum(widths)
        out = torch.empty((M, total), dtype=torch.float32, device=dev)
        st = cur_stream(dev)
        saved, meta, col = [feats], [], 0
        for i in range(k):
            (idx, empty, W1, g1, b1, eps1, rm1, rv1, mom1, nbt1, W2, g2, b2, eps2, rm2, rv2, mom2, nbt2) = args[PER * i:PER * i + PER]
            ns = idx.shape[1]
            h1, h2 = W1.shape[0], W2.shape[0]
            n = M * ns
            idx = idx.contiguous()
            em = empty.view(torch.uint8) if empty.dtype == torch.bool else empty.to(torch.uint8).contiguous()
            W1 = W1.float()
            w1x = W1[:, :3].t().contiguous()                   # (3, h1)
            w1f = W1[:, 3:].contiguous()                       # (h1, C)
            P = feats @ w1f.t()                                # (N, h1): layer 1 per SOURCE point
            W2c = W2.contiguous().float()
            g1c, b1c, g2c, b2c = (t.contiguous().float() for t in (g1, b1, g2, b2))

            def stats(slabs, nslab, C, g, b, eps, rm, rv, nbt, mom):
                mean = torch.empty((C,), dtype=torch.float32, device=dev)
                var, invstd = torch.empty_like(mean), torch.empty_like(mean)
                wsb = lib.crb_bn_workspace_bytes(n, C)
                ws, tk = bnrelu._scratch(dev, wsb)
                bnrelu._bn_check(lib.crb_bn_relu_forward_partials(None, n, C, ptr(slabs), nslab, ptr(g), ptr(b), float(eps), 1, None, 0,
                                                                  ptr(mean), ptr(var), ptr(invstd), ptr(rm), ptr(rv), ptr(nbt),
                                                                  float(mom), ptr(ws), wsb, ptr(tk), st),
                                 'crb_bn_relu_forward_partials')
                bnrelu._touch(rm, rv, nbt)
                return mean, invstd
            # pass 0: statistics of y1 (nothing written but the slab sums)
            nslab = int(lib.crb_group_affine_rows_grad_blocks(M, ns))
            slab1 = torch.empty((nslab, 2, h1), dtype=torch.float32, device=dev)
            check(lib.crb_group_affine_rows_stats_stack(B, M, h1, ns, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc), ptr(idx),
                                                        ptr(em), ptr(w1x), None, None, ptr(slab1), st),
                  'crb_group_affine_rows_stats_stack')
            mean1, invstd1 = stats(slab1, nslab, h1, g1c, b1c, eps1, rm1, rv1, nbt1, mom1)
            # pass A: statistics of y2
            nwave = int(lib.crb_sa_mlp2_train_waves(M))
            slab2 = torch.empty((nwave, 2, h2), dtype=torch.float32, device=dev)
            check(lib.crb_sa_mlp2_train_stats(B, M, ns, h1, h2, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc), ptr(idx), ptr(em),
                                              ptr(w1x), ptr(mean1), ptr(invstd1), ptr(g1c), ptr(b1c), ptr(W2c), ptr(slab2), st),
                  'crb_sa_mlp2_train_stats')
            mean2, invstd2 = stats(slab2, nwave, h2, g2c, b2c, eps2, rm2, rv2, nbt2, mom2)
            # pass B: the output
            arg = torch.empty((M, h2), dtype=torch.int32, device=dev)
            ysel = torch.empty((M, h2), dtype=torch.float32, device=dev)
            check(lib.crb_sa_mlp2_train_max(B, M, ns, h1, h2, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc), ptr(idx), ptr(em),
                                            ptr(w1x), ptr(mean1), ptr(invstd1), ptr(g1c), ptr(b1c), ptr(W2c), ptr(mean2),
                                            ptr(invstd2), ptr(g2c), ptr(b2c), ctypes.c_void_p(out.data_ptr() + 4 * col), total,
                                            ptr(arg), ptr(ysel), st), 'crb_sa_mlp2_train_max')
            saved += [idx, em, w1x, w1f, P, W2c, g1c, b1c, g2c, b2c, mean1, invstd1, mean2, invstd2, arg, ysel]
            meta.append((ns, h1, h2, col))
            col += h2
        ctx.meta = (B, M, total, xc, nc, xyz_c, new_c, meta)
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, gz):
        from crbhip import bnrelu
        PER = SAMlp2TrainConcat.PER
        B, M, total, xc, nc, xyz_c, new_c, meta = ctx.meta
        saved = ctx.saved_tensors
        feats = saved[0]
        dev = gz.device
        st = cur_stream(dev)
        gz = gz.contiguous().float()
        gfeat = None
        grads = []
        for i, (ns, h1, h2, col) in enumerate(meta):
            (idx, em, w1x, w1f, P, W2c, g1c, b1c, g2c, b2c, mean1, invstd1, mean2, invstd2, arg, ysel) = saved[1 + 16 * i:17 + 16 * i]
            n = M * ns
            gp = ctypes.c_void_p(gz.data_ptr() + 4 * col)
            # BatchNorm 2: dbeta / dgamma from the M x h2 selected entries
            d2 = torch.empty((2, h2), dtype=torch.float32, device=dev)
            wsb = lib.crb_bn_workspace_bytes(M, h2)
            ws, tk = bnrelu._scratch(dev, wsb)
            bnrelu._bn_check(lib.crb_bn_relu_max_backward_sums(ptr(ysel), gp, total, M, h2, ptr(mean2), ptr(invstd2), ptr(g2c), ptr(b2c),
                                                               ptr(d2[1]), ptr(d2[0]), ptr(ws), wsb, ptr(tk), st),
                             'crb_bn_relu_max_backward_sums')
            # pass C
            gz1 = torch.empty((n, h1), dtype=torch.float32, device=dev)
            d1 = torch.empty((2, h1), dtype=torch.float32, device=dev)
            dW2 = torch.empty((h2, h1), dtype=torch.float32, device=dev)
            wsf = int(lib.crb_sa_mlp2_train_backward_workspace_floats(M, h1, h2))
            wsp = torch.empty((wsf,), dtype=torch.float32, device=dev)
            check(lib.crb_sa_mlp2_train_backward(B, M, ns, h1, h2, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc), ptr(idx), ptr(em),
                                                 ptr(w1x), ptr(mean1), ptr(invstd1), ptr(g1c), ptr(b1c), ptr(W2c), ptr(mean2),
                                                 ptr(invstd2), ptr(g2c), ptr(b2c), gp, total, ptr(arg), ptr(d2[0]), ptr(d2[1]),
                                                 ptr(gz1), ptr(d1), ptr(dW2), ptr(wsp), wsf, st), 'crb_sa_mlp2_train_backward')
            # pass D: BatchNorm 1 backward inside the scatter kernel of the first layer. Large layers (the RoI-grid scales) scatter in
            # source-row order: a stable device sort of the pairs, then runs of equal rows are added in LDS and cost one row of
            # atomics per 16-pair segment
            gP = torch.zeros((feats.shape[0], h1), dtype=torch.float32, device=dev)
            part = torch.empty((int(lib.crb_group_affine_rows_grad_blocks(M, ns)), 3, h1), dtype=torch.float32, device=dev)
            sp = sr = None
            n_src = feats.shape[0]
            if SORTED_SCATTER and n >= SORTED_SCATTER_MIN_PAIRS:
                sp = torch.empty((n,), dtype=torch.int32, device=dev)
                sr = torch.empty((n,), dtype=torch.int32, device=dev)
                wsb = int(lib.crb_pair_sort_workspace_bytes(M, ns))
                wss = torch.empty((wsb,), dtype=torch.uint8, device=dev)
                check(lib.crb_pair_sort_by_source(B, M, ns, ptr(xc), ptr(nc), ptr(idx), ptr(em), n_src, ptr(sp), ptr(sr), ptr(wss), wsb, st),
                      'crb_pair_sort_by_source')
            if torch.are_deterministic_algorithms_enabled():
                # float atomics add in arrival order; 64-bit fixed point does not care: round(value * 2^40 / R) with R a power of two
                # >= max |grad| * max |gamma invstd| (2^-40 R per addend, room for sums up to 2^22 R)
                R = float(gz1.abs().max()) * float((g1c * invstd1).abs().max())
                scale = 2.0 ** (40 - (math.frexp(R)[1] if R > 0.0 and math.isfinite(R) else 0))
                gP64 = torch.zeros((feats.shape[0], h1), dtype=torch.int64, device=dev)
                check(lib.crb_group_affine_rows_grad_bn_recompute_stack_fixed(B, M, h1, ns, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc),
                                                                              ptr(idx), ptr(em), ptr(w1x), ptr(gz1), ptr(mean1),
                                                                              ptr(invstd1), ptr(g1c), ptr(b1c), ptr(d1[0]), ptr(d1[1]),
                                                                              ptr(sp), ptr(sr), n_src, ptr(gP64), scale, ptr(part), st),
                      'crb_group_affine_rows_grad_bn_recompute_stack_fixed')
                gP = (gP64.double() * (1.0 / scale)).float()
                del gP64
            else:
                check(lib.crb_group_affine_rows_grad_bn_recompute_stack(B, M, h1, ns, ptr(xyz_c), ptr(xc), ptr(P), ptr(new_c), ptr(nc),
                                                                        ptr(idx), ptr(em), ptr(w1x), ptr(gz1), ptr(mean1), ptr(invstd1),
                                                                        ptr(g1c), ptr(b1c), ptr(d1[0]), ptr(d1[1]), ptr(sp), ptr(sr), n_src,
                                                                        ptr(gP), ptr(part), st),
                      'crb_group_affine_rows_grad_bn_recompute_stack')
            del gz1
            if ctx.needs_input_grad[4]:
                gf = gP @ w1f
                gfeat = gf if gfeat is None else gfeat + gf
            gW1 = torch.cat([part.sum(0).t(), _tall_t_matmul(gP, feats)], dim=1)           # (h1, 3+C)
            grads += [None, None, gW1, d1[1], d1[0], None, None, None, None, None,
                      dW2, d2[1], d2[0], None, None, None, None, None]
        return (None, None, None, None, gfeat) + tuple(grads)


def sa_mlp2_train_concat(groupers, mlps, balls, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features):
    """all scales of a StackSAModuleMSG in training mode through SAMlp2TrainConcat; mlps[i] = Sequential(Conv2d 1x1 (no bias),
    BatchNorm2d, ReLU, Conv2d 1x1 (no bias), BatchNorm2d, ReLU) with both BatchNorms in training mode -> (M, sum h2)"""
    from crbhip import bnrelu
    args = []
    for grouper, mlp, ball in zip(groupers, mlps, balls):
        idx, empty = ball if ball is not None else ball_query(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt, new_xyz,
                                                                new_xyz_batch_cnt)
        m = list(mlp)
        for conv, bn in ((m[0], m[1]), (m[3], m[4])):
            if conv is m[0]:
                args += [idx, empty]
            args += [conv.weight.flatten(1), bn.weight, bn.bias, bn.eps, bn.running_mean, bn.running_var, float(bn.momentum),
                     bnrelu._counter(bn)]
    return SAMlp2TrainConcat.apply(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, *args)


def sa_mlp2_max_supported(h1, h2):
    return bool(lib.crb_sa_mlp2_max_supported(int(h1), int(h2)))


@torch.no_grad()
def sa_mlp2_max(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, w1x, w1f_t, b1, w2t, b2, out,
                ball=None):
    """inference-only ball query -> group -> relu(W1 . + b1) -> relu(W2 . + b2) -> max over samples, written into `out`
    ((M, h2) view, may be a column slice of a wider row-major buffer). Operands are the BN-folded 1x1 conv weights of one
    StackSAModuleMSG scale (pointnet2_modules.py:73-112): w1x (3,h1), w1f_t (C,h1), b1 (h1), w2t (h1,h2), b2 (h2)."""
    require_cuda(xyz, new_xyz, features, out)
    idx, empty = ball if ball is not None else ball_query(radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
    h1, h2 = w2t.shape
    M = new_xyz.shape[0]
    assert out.shape == (M, h2) and out.stride(1) == 1
    P = features.contiguous() @ w1f_t                              # (N, h1): layer 1 per SOURCE point, not per pair
    xc, nc = _i32(xyz_batch_cnt), _i32(new_xyz_batch_cnt)
    em = empty.view(torch.uint8) if empty.dtype == torch.bool else empty.to(torch.uint8)
    check(lib.crb_sa_mlp2_max_stack(len(xyz_batch_cnt), M, int(nsample), h1, h2, ptr(xyz.contiguous()), ptr(xc), ptr(P),
                                    ptr(new_xyz.contiguous()), ptr(nc), ptr(idx), ptr(em), ptr(w1x), ptr(b1), ptr(w2t),
                                    ptr(b2), ctypes.c_void_p(out.data_ptr()), out.stride(0), cur_stream(xyz.device)),
          'crb_sa_mlp2_max_stack')
    return out


class FarthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32"""
        require_cuda(xyz)
        assert xyz.is_contiguous()
        B, N, _ = xyz.shape
        out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
        temp = torch.empty((B, N), dtype=torch.float32, device=xyz.device) if N > 40960 else None
        check(lib.crb_farthest_point_sample(B, N, int(npoint), ptr(xyz), ptr(temp), ptr(out), cur_stream(xyz.device)),
              'crb_farthest_point_sample')
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, a=None):
        return None, None


farthest_point_sample = furthest_point_sample = FarthestPointSampling.apply


def stack_farthest_point_sample(xyz, xyz_batch_cnt, npoint):
    """stacked xyz (N,3) with per-frame counts -> concatenated LOCAL indices (sum npoint) (pointnet2_utils.py:187-221).
    Frames are sampled one launch each (frames of different length cannot share the register-resident kernel)."""
    B = len(xyz_batch_cnt)
    cnt = [int(v) for v in xyz_batch_cnt.tolist()]
    if not isinstance(npoint, (list, tuple, torch.Tensor)):
        npoint = [npoint] * B
    npoint = [int(v) for v in (npoint.tolist() if isinstance(npoint, torch.Tensor) else npoint)]
    outs, s = [], 0
    for b in range(B):
        outs.append(farthest_point_sample(xyz[s:s + cnt[b]].unsqueeze(0).contiguous(), npoint[b]).view(-1))
        s += cnt[b]
    return torch.cat(outs)


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        require_cuda(unknown, known)
        assert unknown.dim() == 2 and unknown.shape[1] == 3 and known.dim() == 2 and known.shape[1] == 3
        N = unknown.shape[0]
        dist2 = torch.empty((N, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((N, 3), dtype=torch.int32, device=unknown.device)
        check(lib.crb_three_nn_stack(len(unknown_batch_cnt), N, ptr(unknown.contiguous()), ptr(_i32(unknown_batch_cnt)),
                                     ptr(known.contiguous()), ptr(_i32(known_batch_cnt)), ptr(dist2), ptr(idx),
                                     cur_stream(unknown.device)), 'crb_three_nn_stack')
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (M,C), idx (N,3), weight (N,3) -> (N,C)"""
        require_cuda(features, idx, weight)
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        f, i, w = features.contiguous().float(), idx.contiguous(), weight.contiguous().float()
        ctx.three_interpolate_for_backward = (i, w, f.shape[0])
        out = torch.empty((i.shape[0], f.shape[1]), dtype=torch.float32, device=f.device)
        check(lib.crb_three_interpolate_stack(i.shape[0], f.shape[1], ptr(f), ptr(i), ptr(w), ptr(out),
                                              cur_stream(f.device)), 'crb_three_interpolate_stack')
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, M = ctx.three_interpolate_for_backward
        g = grad_out.contiguous().float()
        grad_features = torch.zeros((M, g.shape[1]), dtype=torch.float32, device=g.device)
        check(lib.crb_three_interpolate_grad_stack(g.shape[0], g.shape[1], ptr(g), ptr(idx), ptr(weight),
                                                   ptr(grad_features), cur_stream(g.device)),
              'crb_three_interpolate_grad_stack')
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply
