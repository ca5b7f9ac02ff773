"""Detection losses with the reference's class names (pcdet/utils/loss_utils.py:9-232).
Device placement follows the input tensors instead of hard-coded .cuda() calls."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import box_utils


class SigmoidFocalClassificationLoss(nn.Module):
    def __init__(self, gamma: float = 2.0, alpha: float = 0.25):
        super().__init__()
        self.alpha, self.gamma = alpha, gamma

    @staticmethod
    def sigmoid_cross_entropy_with_logits(input, target):
        return torch.clamp(input, min=0) - input * target + torch.log1p(torch.exp(-torch.abs(input)))

    def forward(self, input, target, weights):
        p = torch.sigmoid(input)
        alpha_weight = target * self.alpha + (1 - target) * (1 - self.alpha)
        pt = target * (1.0 - p) + (1.0 - target) * p
        loss = alpha_weight * torch.pow(pt, self.gamma) * self.sigmoid_cross_entropy_with_logits(input, target)
        if weights.dim() == 2 or (weights.dim() == 1 and target.dim() == 2):
            weights = weights.unsqueeze(-1)
        assert weights.dim() == loss.dim()
        return loss * weights


class WeightedSmoothL1Loss(nn.Module):
    def __init__(self, beta: float = 1.0 / 9.0, code_weights: list = None):
        super().__init__()
        self.beta = beta
        if code_weights is not None:
            self.register_buffer('code_weights', torch.tensor(np.array(code_weights, dtype=np.float32)),
                                 persistent=False)
        else:
            self.code_weights = None

    @staticmethod
    def smooth_l1_loss(diff, beta):
        if beta < 1e-5:
            return torch.abs(diff)
        n = torch.abs(diff)
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights.to(diff.device).view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


class WeightedCrossEntropyLoss(nn.Module):
    def forward(self, input, target, weights):
        """input (B,A,C) logits, target (B,A,C) one-hot, weights (B,A) -> (B,A)"""
        input = input.permute(0, 2, 1)
        target = target.argmax(dim=-1)
        return F.cross_entropy(input, target, reduction='none') * weights


def get_corner_loss_lidar(pred_bbox3d, gt_bbox3d):
    """(N,7),(N,7) -> (N) huber corner distance, min over the heading-flipped gt (loss_utils.py:209-232)"""
    assert pred_bbox3d.shape[0] == gt_bbox3d.shape[0]
    pred = box_utils.boxes_to_corners_3d(pred_bbox3d)
    gt = box_utils.boxes_to_corners_3d(gt_bbox3d)
    gt_flip_box = gt_bbox3d.clone()
    gt_flip_box[:, 6] += np.pi
    gt_flip = box_utils.boxes_to_corners_3d(gt_flip_box)
    dist = torch.min(torch.norm(pred - gt, dim=2), torch.norm(pred - gt_flip, dim=2))
    return WeightedSmoothL1Loss.smooth_l1_loss(dist, beta=1.0).mean(dim=1)
