"""ORACLE — test infrastructure only: CPU port of one SECOND forward+backward step, used as the `cpu_baseline` leg of
bench.py and by smoke()/tests as a checker. The sparse half runs on the C restatements of this directory
(voxelize_oracle.c, sparse_conv_oracle.c) wrapped as autograd Functions; the dense half (BEV backbone, anchor head,
target assignment, losses) is oracle/anchor_head_oracle.py — functional torch on the CPU, pinned by the reference's goldens. Follows pcdet/models/detectors/second_net.py:9-34 and the layer
list of pcdet/models/backbones_3d/spconv_backbone.py:77-117."""
import numpy as np
import torch
import torch.nn.functional as F

import oracle


class _OracleConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, nbr, n_in):
        ctx.nbr, ctx.n_in = nbr, n_in
        ctx.save_for_backward(x, w)
        return torch.from_numpy(oracle.conv_fwd(x.detach().numpy(), w.detach().numpy(), nbr))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyn = np.ascontiguousarray(dy.numpy())
        dx = torch.from_numpy(oracle.conv_dgrad(dyn, w.detach().numpy(), ctx.nbr, ctx.n_in))
        dw = torch.from_numpy(oracle.conv_wgrad(x.detach().numpy(), dyn, ctx.nbr, w.shape[0]))
        return dx, dw, None, None


def _bn_relu(x, bn):
    """train-mode BatchNorm1d over active voxels + ReLU with the layer's affine parameters"""
    y = F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
    return torch.relu(y)


def sparse_backbone_cpu(backbone, feats, coords, batch_size):
    """backbone: a pcdet VoxelBackBone8x whose parameters live on the CPU. feats (N,C) tensor, coords (N,4) int32 ndarray.
    -> dense BEV tensor (B, 256, H, W) and the per-level coordinates"""
    shape = list(backbone.sparse_shape)
    cache = {}
    x = feats
    levels = []

    def run_block(seq, x, coords, shape):
        conv, bn = seq[0], seq[1]
        key = conv.indice_key
        if conv.subm:
            if key not in cache:
                cache[key] = oracle.subm_nbr(coords, shape, conv.kernel_size)
            nbr, n_in = cache[key], len(coords)
        else:
            oc, oshape = oracle.spconv_out(coords, shape, conv.kernel_size, conv.stride, conv.padding)
            nbr = oracle.spconv_nbr(coords, shape, oc, conv.kernel_size, conv.stride, conv.padding)
            n_in = len(coords)
            coords, shape = oc, oshape
        y = _OracleConv.apply(x, conv.weight_kio(), nbr, n_in)
        return _bn_relu(y, bn), coords, shape

    x, coords, shape = run_block(backbone.conv_input, x, coords, shape)
    for stage in (backbone.conv1, backbone.conv2, backbone.conv3, backbone.conv4):
        for blk in stage:
            x, coords, shape = run_block(blk, x, coords, shape)
        levels.append((coords, shape))
    x, coords, shape = run_block(backbone.conv_out, x, coords, shape)
    c = torch.from_numpy(coords).long()
    dense = torch.zeros(batch_size, shape[0], shape[1], shape[2], x.shape[1])
    dense = dense.index_put((c[:, 0], c[:, 1], c[:, 2], c[:, 3]), x)
    dense = dense.permute(0, 4, 1, 2, 3).contiguous()
    B, C, D, H, W = dense.shape
    return dense.view(B, C * D, H, W), levels


def second_step_cpu(model, points, frame_offsets, gt_boxes, max_voxels=16000, max_points=5, return_parts=False):
    """one fwd+bwd of SECOND on the CPU. model: pcdet SECONDNet on the CPU in train() mode — used as a PARAMETER CONTAINER
    only: the sparse half runs on the C restatements, the dense half (BEV backbone, head, anchors, target assignment,
    losses) on oracle/anchor_head_oracle.py, none of the product's forward code. Gradients land in the parameters' .grad.
    points (n,C) ndarray, frame_offsets (B+1), gt_boxes (B,G,8) ndarray -> loss (float) [, (cls, loc, dir) floats]"""
    from oracle import anchor_head_oracle as aho
    ds = model.dataset
    B = len(frame_offsets) - 1
    v, c, n, _ = oracle.voxelize_batch(points, frame_offsets, ds.point_cloud_range[:3], ds.voxel_size,
                                       [int(g) for g in ds.grid_size], max_voxels, max_points)
    feats = torch.from_numpy(oracle.mean_vfe(v, n))
    bev, _ = sparse_backbone_cpu(model.backbone_3d, feats, c, B)
    feats2d = aho.bev_backbone(model.backbone_2d, bev)
    cls, box, dr = aho.head_preds(model.dense_head, feats2d)
    hcfg = model.model_cfg.DENSE_HEAD
    acfg = [dict(a) for a in hcfg.ANCHOR_GENERATOR_CONFIG]
    fm = [np.asarray(ds.grid_size[:2]) // a['feature_map_stride'] for a in acfg]
    anchors = aho.generate_anchors(ds.point_cloud_range, acfg, fm)
    labels, targets, _ = aho.assign_targets(anchors, torch.from_numpy(gt_boxes), list(ds.class_names), acfg)
    lw = hcfg.LOSS_CONFIG.LOSS_WEIGHTS
    loss, lc, ll, ld = aho.rpn_loss(cls, box, dr, labels, targets, anchors, num_class=len(ds.class_names),
                                    dir_offset=hcfg.DIR_OFFSET, num_bins=hcfg.NUM_DIR_BINS, w_cls=lw['cls_weight'],
                                    w_loc=lw['loc_weight'], w_dir=lw['dir_weight'])
    model.zero_grad(set_to_none=True)
    loss.backward()
    if return_parts:
        return float(loss.detach()), (float(lc.detach()), float(ll.detach()), float(ld.detach()))
    return float(loss.detach())
