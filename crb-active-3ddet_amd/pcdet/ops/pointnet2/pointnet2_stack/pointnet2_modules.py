"""StackSAModuleMSG / build_local_aggregation_module (pointnet2_modules.py:10-112): multi-scale set abstraction on
stacked batches: ball query + grouping (HIP) -> shared 1x1 MLP (+BN+ReLU) -> max over the neighbourhood."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from crbhip import bnrelu

from . import pointnet2_utils
from ....utils.fold_utils import fold_conv_bn
from ....utils.linear_rows import LinearRows as _LinearRows


def _split_first_layer(w, b):
    """(h1, 3+C, 1, 1) -> W1x (3, h1), W1f^T (C, h1), b1: the operand layout of crb_sa_mlp2_max_stack"""
    w = w.flatten(1)
    return w[:, :3].t().contiguous(), w[:, 3:].t().contiguous(), b.contiguous()


def _transpose_second_layer(w, b):
    return w.flatten(1).t().contiguous(), b.contiguous()


FUSED_SA_EVAL = True   # inference: group + 2-layer MLP + max in one HIP kernel (crb_sa_mlp2_max_stack)
ROWS_TRAIN = True      # training: row-major grouped matrix -> GEMM + fused BN/ReLU row kernels -> max (no MIOpen BN2d / transposes)
SPLIT_FIRST_LAYER = True   # training rows path: layer 1 = gather of per-source-point products + offset term (no grouped matrix)
FUSED_BN_MAX = True        # training rows path: last BatchNorm+ReLU, max over nsample and the concat of the scales in one op
FUSED_FIRST_BN = True      # training rows path: first conv + BatchNorm + ReLU as one node, BN backward inside the gradient kernel
FUSED_GROUP = True     # one HIP launch builds the (1, 3+C, M, ns) MLP input (False: QueryAndGroup + permute copy)
# training, two-layer MLPs: the whole module as one autograd node that keeps no (M*ns, H) activation (csrc/sa_mlp_train.hip:
# statistics passes recompute the layers from the ball-query indices); CRB_SA_TRAIN_FUSED=0 = the rows path below (A/B)
FUSED_TRAIN = __import__('os').environ.get('CRB_SA_TRAIN_FUSED', '1') == '1'


def build_local_aggregation_module(input_channels, config):
    name = config.get('NAME', 'StackSAModuleMSG')
    if name != 'StackSAModuleMSG':
        raise NotImplementedError(name + ' (VectorPool is PV-RCNN++ only: out of scope, SURVEY §2.1 row 4)')
    mlps = [[input_channels] + list(m) for m in config.MLPS]
    layer = StackSAModuleMSG(radii=config.POOL_RADIUS, nsamples=config.NSAMPLE, mlps=mlps, use_xyz=True,
                             pool_method='max_pool')
    return layer, sum(m[-1] for m in mlps)


class StackSAModuleMSG(nn.Module):
    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            layers = []
            for c_in, c_out in zip(spec[:-1], spec[1:]):
                layers += [nn.Conv2d(c_in, c_out, kernel_size=1, bias=False), nn.BatchNorm2d(c_out), nn.ReLU()]
            self.mlps.append(nn.Sequential(*layers))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _mlp_eval_folded(mlp, x):
        """inference: BatchNorm2d folded into the preceding 1x1 conv (w' = w * gamma/sqrt(var+eps), b' = beta - mean*that);
        the grouped tensors are GB-sized (RoI grid: (1,131,442k,16)), so the separate BN pass was the single largest
        item of the CRB scoring profile"""
        mods = list(mlp)
        i = 0
        while i < len(mods):
            w, b = fold_conv_bn(mods[i], mods[i + 1])
            x = F.relu(F.conv2d(x, w, b), inplace=True)
            i += 3
        return x

    @staticmethod
    def _folded_pair(mlp):
        """kernel operands of a Conv-BN-ReLU-Conv-BN-ReLU stack with eval BN folded (cached on the convs), or None"""
        mods = list(mlp)
        if len(mods) != 6:
            return None
        for conv, bn in ((mods[0], mods[1]), (mods[3], mods[4])):
            if not isinstance(conv, nn.Conv2d) or not isinstance(bn, nn.BatchNorm2d) or bn.training:
                return None
        w1x, w1f_t, b1 = fold_conv_bn(mods[0], mods[1], _split_first_layer)
        w2t, b2 = fold_conv_bn(mods[3], mods[4], _transpose_second_layer)
        return w1x, w1f_t, b1, w2t, b2

    def _forward_fused_eval(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, query_group=None):
        folded = [self._folded_pair(m) for m in self.mlps]
        if any(f is None or not pointnet2_utils.sa_mlp2_max_supported(f[3].shape[0], f[3].shape[1]) for f in folded):
            return None
        widths = [f[3].shape[1] for f in folded]
        out = torch.empty((new_xyz.shape[0], sum(widths)), dtype=torch.float32, device=xyz.device)
        col = 0
        balls = self._balls(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, query_group)
        for grouper, f, w, ball in zip(self.groupers, folded, widths, balls):
            pointnet2_utils.sa_mlp2_max(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt, new_xyz,
                                        new_xyz_batch_cnt, features, *f, out[:, col:col + w], ball=ball)
            col += w
        return out

    def _balls(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, query_group=None):
        """ball queries of all scales; two scales share one scan of the points (crb_ball_query2_stack)"""
        gs = list(self.groupers)
        if len(gs) == 2:
            return pointnet2_utils.ball_query_pair(gs[0].radius, gs[0].nsample, gs[1].radius, gs[1].nsample, xyz,
                                                   xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, group=query_group)
        return [None] * len(gs)

    def _train_fused_ok(self):
        """every scale is Conv(1x1, no bias)-BN-ReLU-Conv(1x1, no bias)-BN-ReLU with train-mode BatchNorms and a shape the
        recompute kernels have (widths 16 / 32 / 64, nsample a multiple of 16); per-frame statistics (batched CRB stage 2) keep
        the rows path"""
        if bnrelu.frame_groups_active():
            return False
        for grouper, mlp in zip(self.groupers, self.mlps):
            mods = list(mlp)
            if len(mods) != 6 or mods[0].bias is not None or mods[3].bias is not None:
                return False
            for bn in (mods[1], mods[4]):
                if not (bn.training and bn.momentum is not None and bn.affine and bn.track_running_stats and
                        bn.running_mean.is_contiguous() and bn.running_var.is_contiguous()):
                    return False
            if not pointnet2_utils.sa_mlp2_train_supported(mods[0].out_channels, mods[3].out_channels, grouper.nsample):
                return False
        return True

    @staticmethod
    def _rows_ok(mlp, features):
        mods = list(mlp)
        if len(mods) % 3:
            return False
        x = features.new_empty((2, 4))
        for i in range(0, len(mods), 3):
            conv, bn, act = mods[i], mods[i + 1], mods[i + 2]
            if not (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and isinstance(bn, nn.BatchNorm2d)
                    and isinstance(act, nn.ReLU)):
                return False
            if not bnrelu.supported(x.new_empty((2, conv.out_channels)), bn):
                return False
        return True

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True,
                query_group=None):
        """xyz (N,3), features (N,C), new_xyz (M,3) -> new_xyz, new_features (M, sum C_out). query_group (not in the reference
        signature): the caller's promise that new_xyz comes in spatially compact groups of that many consecutive rows inside
        one frame; only lets the ball query prefilter per group, the result does not depend on it."""
        if FUSED_SA_EVAL and not self.training and not torch.is_grad_enabled() and features is not None \
                and xyz.is_cuda and self.pool_method == 'max_pool' and all(g.use_xyz for g in self.groupers):
            fused = self._forward_fused_eval(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features, query_group)
            if fused is not None:
                return new_xyz, fused
        outs = []
        if ROWS_TRAIN and features is not None and xyz.is_cuda and self.pool_method == 'max_pool' \
                and all(g.use_xyz for g in self.groupers) and all(self._rows_ok(m, features) for m in self.mlps):
            M = new_xyz.shape[0]
            balls = self._balls(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, query_group)
            if FUSED_TRAIN and self.training and M > 0 and self._train_fused_ok():
                return new_xyz, pointnet2_utils.sa_mlp2_train_concat(self.groupers, self.mlps, balls, xyz, xyz_batch_cnt, new_xyz,
                                                                     new_xyz_batch_cnt, features)
            fuse_max = FUSED_BN_MAX and self.training and M > 0 and \
                all(list(m)[-2].training and list(m)[-2].momentum is not None for m in self.mlps)
            for grouper, mlp, ball in zip(self.groupers, self.mlps, balls):
                mods = list(mlp)
                split = SPLIT_FIRST_LAYER and mods[0].bias is None and mods[0].out_channels in (16, 32, 64, 128)
                if not split:
                    x, _ = pointnet2_utils.query_and_group_rows(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt,
                                                                new_xyz, new_xyz_batch_cnt, features, ball=ball)   # (M*ns, 3+C)
                for i in range(0, len(mods), 3):
                    conv, bn = mods[i], mods[i + 1]
                    if i == 0 and split and FUSED_FIRST_BN and (i + 3 < len(mods) or not fuse_max) and bn.training and \
                            bn.momentum is not None and M > 0 and not bnrelu.frame_groups_active() and \
                            bn.running_mean.is_contiguous() and bn.running_var.is_contiguous():
                        # first conv + its BatchNorm + ReLU as one autograd node: the BatchNorm backward is applied inside the
                        # gather-scatter gradient kernel (no (M*ns, H) gradient of the conv output in HBM)
                        x = pointnet2_utils.grouped_first_layer_bn_relu(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt,
                                                                        new_xyz, new_xyz_batch_cnt, features,
                                                                        conv.weight.flatten(1), bn, ball=ball)
                        continue
                    if i == 0 and split:
                        x = pointnet2_utils.grouped_first_layer_rows(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt,
                                                                     new_xyz, new_xyz_batch_cnt, features,
                                                                     conv.weight.flatten(1), ball=ball)    # (M*ns, H)
                    else:
                        x = _LinearRows.apply(x, conv.weight.flatten(1))
                    if conv.bias is not None:
                        x = x + conv.bias
                    if i + 3 < len(mods) or not fuse_max:
                        x = bnrelu.bn_relu(x, bn, relu=True)
                if fuse_max:
                    outs.append(x)                                         # pre-BN rows of the last layer
                else:
                    outs.append(x.view(M, grouper.nsample, x.shape[1]).max(dim=1).values)
            if fuse_max:
                return new_xyz, bnrelu.bn_relu_max_concat(outs, [g.nsample for g in self.groupers],
                                                          [list(m)[-2] for m in self.mlps])
            return new_xyz, torch.cat(outs, dim=1)
        for grouper, mlp in zip(self.groupers, self.mlps):
            if FUSED_GROUP and features is not None and grouper.use_xyz and xyz.is_cuda:
                x_in, _ = pointnet2_utils.query_and_group_fused(grouper.radius, grouper.nsample, xyz, xyz_batch_cnt,
                                                                new_xyz, new_xyz_batch_cnt, features)   # (1, 3+C, M, ns)
            else:
                grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)  # (M, C, ns)
                x_in = grouped.permute(1, 0, 2).unsqueeze(0)
            if not self.training and not torch.is_grad_enabled():
                x = self._mlp_eval_folded(mlp, x_in)
            else:
                x = mlp(x_in)                                                                   # (1, C', M, ns)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
            else:
                raise NotImplementedError
            outs.append(x.squeeze(0).permute(1, 0))
        return new_xyz, torch.cat(outs, dim=1)
