"""Pointwise (1x1) convolutions as row GEMMs.

A channels_last (N,C,H,W) activation is a row-major (N*H*W, C) matrix, so a 1x1 convolution is `rows @ W^T` with no layout
change. The reference runs these layers as nn.Conv2d / nn.ConvTranspose2d (anchor_head_single.py:16-33 for the cls / box / dir
heads, base_bev_backbone.py:52-58 for the stride-1 up-sampling branch); MIOpen picks implicit-GEMM / CK backward-data kernels
for them that reach 40-55 TFLOP/s on the 563k-row BEV maps, a plain GEMM does not need the convolution machinery."""
import torch


class LinearRows(torch.autograd.Function):
    """y = x @ w^T (+ bias) for a tall (rows, Cin) matrix. The weight gradient dy^T x reduces over 10^5..10^7 rows into a small
    (Cout, Cin) tile: as one GEMM it gets a handful of workgroups (3 ms per call at the RoI-grid shape of PV-RCNN); here the
    rows are cut into 256 slices multiplied as one batched GEMM and summed."""
    SLICES = 256

    @staticmethod
    def forward(ctx, x, w, bias=None):
        ctx.save_for_backward(x, w)
        return x @ w.t() if bias is None else torch.addmm(bias, x, w.t())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            R, S = x.shape[0], LinearRows.SLICES
            if R >= 64 * S:
                r0 = (R // S) * S
                dw = torch.bmm(dy[:r0].view(S, r0 // S, -1).transpose(1, 2), x[:r0].view(S, r0 // S, -1)).sum(0)
                if r0 < R:
                    dw = dw + dy[r0:].t() @ x[r0:]
            else:
                dw = dy.t() @ x
        db = dy.sum(0) if len(ctx.needs_input_grad) > 2 and ctx.needs_input_grad[2] else None
        return dx, dw, db


def tall_t_matmul(a, b, slices=LinearRows.SLICES):
    """a^T @ b for two tall matrices a (R, P), b (R, Q) -> (P, Q): the reduction over the rows cut into `slices` slices multiplied
    as one batched GEMM and summed (as one GEMM a 32 x 32 x 277,496 product gets one workgroup: 0.57 ms, 1 % of the f32 MFMA peak -
    the per-source-point weight gradients of the set-abstraction layers were 3.4 ms of a PV-RCNN step, tools/prof_gemms.py)"""
    R = a.shape[0]
    if R < 64 * slices:
        return a.t() @ b
    r0 = (R // slices) * slices
    out = torch.bmm(a[:r0].view(slices, r0 // slices, -1).transpose(1, 2), b[:r0].view(slices, r0 // slices, -1)).sum(0)
    if r0 < R:
        out = out + a[r0:].t() @ b[r0:]
    return out


def rows_view(x):
    """(N,C,H,W) channels_last -> (N*H*W, C) view of the same storage, or None when x is not laid out that way"""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        return None
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c)


def rows_to_nchw(rows, n, h, w):
    """(N*H*W, C) rows -> (N,C,H,W) channels_last view"""
    return rows.view(n, h, w, rows.shape[1]).permute(0, 3, 1, 2)
