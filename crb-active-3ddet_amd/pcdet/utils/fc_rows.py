"""nn.Sequential stacks of Linear -> BatchNorm1d -> ReLU on (N, C) rows (keypoint feature fusion, voxel_set_abstraction.py:150-154;
PointHeadTemplate.make_fc_layers, point_head_template.py:36-47) with the BatchNorm + ReLU pairs on the fused row kernels
(crbhip.bnrelu: statistics + apply, 2 launches forward and 2 backward) instead of torch's native batch norm, whose kernels take
125 / 148 us forward / backward on the (32768, 256) keypoint rows against ~15 / 25 us. Same parameters, same state_dict, same
running-statistics updates; anything the row kernels do not cover runs the modules as they are."""
import torch.nn as nn
import torch.nn.functional as F

from crbhip import bnrelu

FUSED_FC_BN = True


def fc_rows(seq, x):
    """seq: nn.Sequential of Linear / BatchNorm1d / ReLU / ...; x (N, C) -> seq(x)"""
    if not (FUSED_FC_BN and x.is_cuda and x.dim() == 2) or bnrelu.frame_groups_active():
        return seq(x)                       # batched CRB stage 2 keeps its per-frame BatchNorm1d forward (pcdet/utils/frame_bn.py)
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) \
                and isinstance(mods[i + 2], nn.ReLU):
            y = F.linear(x, m.weight, m.bias)
            if bnrelu.supported(y, mods[i + 1]) and (not mods[i + 1].training or mods[i + 1].momentum is not None):
                x = bnrelu.bn_relu(y, mods[i + 1], relu=True)
                i += 3
                continue
            x = y
            i += 1
            continue
        x = m(x)
        i += 1
    return x
