"""ProposalTargetLayer (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:8-228): RoI sampling for the
second stage, batched over frames and free of host synchronisation.

Same sampling rule as the reference (fg >= min(REG_FG_THRESH, CLS_FG_THRESH), hard bg in [CLS_BG_THRESH_LO, REG_FG_THRESH),
easy bg below, FG_RATIO / HARD_BG_RATIO quotas, fg without replacement, bg with replacement); the random draws come
from a torch.Generator (device RNG) or from injected uniforms instead of np.random / CPU torch.randint
(proposal_target_layer.py:134,171,175) — the reference's CPU RNG stream is not reproduced."""
import numpy as np
import torch
import torch.nn as nn

from ....ops.iou3d_nms import iou3d_nms_utils


# the whole layer after the IoU matrix as one launch (crbhip.rcnn_loss.roi_sample_targets: ~135 elementwise / sort / gather launches of
# a PV-RCNN step); CRB_ROI_SAMPLER_FUSED=0 = the torch expressions below (A/B, and what the kernel is tested against)
FUSED_SAMPLER = __import__('os').environ.get('CRB_ROI_SAMPLER_FUSED', '1') == '1'


class ProposalTargetLayer(nn.Module):
    def __init__(self, roi_sampler_cfg):
        super().__init__()
        self.roi_sampler_cfg = roi_sampler_cfg
        self.generator = None          # optional torch.Generator for reproducible draws
        # parity tests against a reference run whose np.random / torch.randint draws were recorded: use these RoIs instead of
        # drawing. injected_indices (B, ROI_PER_IMAGE) long = positions in the proposal list; injected_rois (B, ROI_PER_IMAGE, 7)
        # = the boxes themselves, looked up in the proposal list (proposals with EQUAL scores have no defined order: topk)
        self.injected_indices = None
        self.injected_rois = None

    def _fused_cfg(self):
        cfg = self.roi_sampler_cfg
        if cfg.CLS_SCORE_TYPE not in ('roi_iou', 'cls'):
            return None
        key = (int(cfg.ROI_PER_IMAGE), float(cfg.FG_RATIO), bool(cfg.get('SAMPLE_ROI_BY_EACH_CLASS', False)), cfg.CLS_SCORE_TYPE,
               float(cfg.REG_FG_THRESH), float(cfg.CLS_FG_THRESH), float(cfg.CLS_BG_THRESH), float(cfg.CLS_BG_THRESH_LO),
               float(cfg.HARD_BG_RATIO))
        hit = self.__dict__.get('_crb_sampler_cfg')
        if hit is None or hit[0] != key:
            from crbhip import rcnn_loss
            c = rcnn_loss.RoiSamplerCfg(key[0], int(np.round(key[1] * key[0])), int(key[2]), 0 if key[3] == 'roi_iou' else 1,
                                        min(key[4], key[5]), key[4], key[5], key[6], key[7], key[8], key[5] - key[6])
            hit = self.__dict__['_crb_sampler_cfg'] = (key, c)
        return hit[1]

    def forward_fused(self, batch_dict, uniforms=None):
        """the layer as one launch behind the IoU matrix; None when this call is not what the kernel implements (injected RoIs of the
        parity tests, more than 1024 proposals per frame, host tensors)"""
        rois, gt_boxes = batch_dict.get('rois', None), batch_dict.get('gt_boxes', None)
        # (an instance whose sampling methods were replaced - the golden tests feed recorded draws that way - keeps the torch layer)
        if rois is None or gt_boxes is None or not rois.is_cuda or 'sample_rois_for_rcnn' in self.__dict__ or \
                'subsample_rois_batched' in self.__dict__:
            return None
        from crbhip import rcnn_loss
        c = self._fused_cfg()
        if c is None or self.injected_rois is not None or self.injected_indices is not None or \
                rois.shape[1] > rcnn_loss.MAX_PROPOSALS or gt_boxes.shape[1] == 0 or gt_boxes.shape[-1] < 8:
            return None
        B, R, G = rois.shape[0], rois.shape[1], gt_boxes.shape[1]
        iou = iou3d_nms_utils.boxes_iou3d_gpu(rois.reshape(B * R, rois.shape[-1])[:, 0:7], gt_boxes.reshape(B * G, -1)[:, 0:7])
        if uniforms is None:
            u_perm = torch.rand((B, R), device=rois.device, generator=self.generator)
            u_slot = torch.rand((B, c.roi_per_image), device=rois.device, generator=self.generator)
        else:
            u_perm, u_slot = uniforms
        out = rcnn_loss.roi_sample_targets(rois, batch_dict['roi_scores'], batch_dict['roi_labels'], gt_boxes, iou, u_perm, u_slot, c)
        out.pop('sampled')
        return out

    def forward(self, batch_dict, uniforms=None):
        if FUSED_SAMPLER:
            out = self.forward_fused(batch_dict, uniforms)
            if out is not None:
                return out
        cfg = self.roi_sampler_cfg
        rois, gt_of_rois, ious, scores, labels = self.sample_rois_for_rcnn(batch_dict, uniforms)
        reg_valid_mask = (ious > cfg.REG_FG_THRESH).long()
        if cfg.CLS_SCORE_TYPE == 'cls':
            cls_labels = (ious > cfg.CLS_FG_THRESH).long()
            ignore = (ious > cfg.CLS_BG_THRESH) & (ious < cfg.CLS_FG_THRESH)
            cls_labels = torch.where(ignore, torch.full_like(cls_labels, -1), cls_labels)
        elif cfg.CLS_SCORE_TYPE == 'roi_iou':
            fg = ious > cfg.CLS_FG_THRESH
            bg = ious < cfg.CLS_BG_THRESH
            soft = (ious - cfg.CLS_BG_THRESH) / (cfg.CLS_FG_THRESH - cfg.CLS_BG_THRESH)
            cls_labels = torch.where(fg, torch.ones_like(ious), torch.where(bg, torch.zeros_like(ious), soft))
        else:
            raise NotImplementedError
        return {'rois': rois, 'gt_of_rois': gt_of_rois, 'gt_iou_of_rois': ious, 'roi_scores': scores,
                'roi_labels': labels, 'reg_valid_mask': reg_valid_mask, 'rcnn_cls_labels': cls_labels}

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def max_iou_with_same_class_batched(rois, roi_labels, gt_boxes):
        """rois (B,R,7), roi_labels (B,R) long, gt_boxes (B,G,8) zero padded -> max_overlaps (B,R), gt_assignment (B,R).
        Batched form of get_max_iou_with_same_class (proposal_target_layer.py:195-228): IoU3D only against ground truths
        of the RoI's own class; RoIs without a same-class gt get overlap 0 and assignment 0."""
        B, R, _ = rois.shape
        G = gt_boxes.shape[1]
        nonzero = gt_boxes.sum(-1) != 0           # the whole row, class label included (reference :93 `cur_gt[k].sum() == 0`)
        idx = torch.arange(G, device=rois.device).view(1, G)
        last = torch.where(nonzero, idx, torch.zeros_like(idx)).max(dim=1, keepdim=True)[0]
        valid = idx <= last
        iou = iou3d_nms_utils.boxes_iou3d_gpu(rois.reshape(B * R, 7)[:, 0:7], gt_boxes.reshape(B * G, -1)[:, 0:7])
        ar = torch.arange(B, device=rois.device)
        iou = iou.view(B, R, B, G)[ar, :, ar]                                               # (B,R,G) block diagonal
        same = (roi_labels[:, :, None] == gt_boxes[:, None, :, -1].long()) & valid[:, None, :]
        iou = torch.where(same, iou, iou.new_full((), -1.0))
        mx, arg = iou.max(dim=2)
        none = mx < 0
        return torch.where(none, torch.zeros_like(mx), mx), torch.where(none, torch.zeros_like(arg), arg)

    def sample_rois_for_rcnn(self, batch_dict, uniforms=None):
        cfg = self.roi_sampler_cfg
        rois, roi_scores, roi_labels, gt_boxes = (batch_dict['rois'], batch_dict['roi_scores'], batch_dict['roi_labels'],
                                                  batch_dict['gt_boxes'])
        B = rois.shape[0]
        if cfg.get('SAMPLE_ROI_BY_EACH_CLASS', False):
            max_overlaps, gt_assignment = self.max_iou_with_same_class_batched(rois, roi_labels, gt_boxes)
        else:
            G = gt_boxes.shape[1]
            iou = iou3d_nms_utils.boxes_iou3d_gpu(rois.reshape(-1, rois.shape[-1])[:, 0:7],
                                                  gt_boxes.reshape(B * G, -1)[:, 0:7])
            iou = iou.view(B, -1, B, G)[torch.arange(B), :, torch.arange(B)]
            max_overlaps, gt_assignment = iou.max(dim=2)
        if self.injected_rois is not None:
            inj = self.injected_rois.to(rois.device, rois.dtype)
            dist = (rois[:, None, :, :7] - inj[:, :, None, :7]).abs().amax(-1)                # (B, P, R)
            best, sampled = dist.min(dim=2)
            assert float(best.max()) < 1e-3, 'an injected RoI is not among the proposals'
        elif self.injected_indices is not None:
            sampled = self.injected_indices.to(max_overlaps.device).long()
            assert sampled.shape == (B, cfg.ROI_PER_IMAGE)
        else:
            sampled = self.subsample_rois_batched(max_overlaps, uniforms)                 # (B, ROI_PER_IMAGE)
        g = lambda t: torch.gather(t, 1, sampled)
        batch_rois = torch.gather(rois, 1, sampled[..., None].expand(-1, -1, rois.shape[-1]))
        assign = g(gt_assignment)
        batch_gt = torch.gather(gt_boxes, 1, assign[..., None].expand(-1, -1, gt_boxes.shape[-1]))
        return batch_rois, batch_gt, g(max_overlaps), g(roi_scores), g(roi_labels)

    def subsample_rois_batched(self, max_overlaps, uniforms=None):
        """max_overlaps (B,R) -> sampled indices (B, ROI_PER_IMAGE), ordered [fg..., hard bg..., easy bg...]"""
        cfg = self.roi_sampler_cfg
        B, R = max_overlaps.shape
        P = cfg.ROI_PER_IMAGE
        dev = max_overlaps.device
        fg_quota = int(np.round(cfg.FG_RATIO * P))
        fg_thresh = min(cfg.REG_FG_THRESH, cfg.CLS_FG_THRESH)
        fg = max_overlaps >= fg_thresh
        easy = max_overlaps < cfg.CLS_BG_THRESH_LO
        hard = (max_overlaps < cfg.REG_FG_THRESH) & (max_overlaps >= cfg.CLS_BG_THRESH_LO)
        if uniforms is None:
            u_perm = torch.rand((B, R), device=dev, generator=self.generator)
            u_slot = torch.rand((B, P), device=dev, generator=self.generator)
        else:
            u_perm, u_slot = uniforms
        n_fg, n_hard, n_easy = fg.sum(1), hard.sum(1), easy.sum(1)
        n_bg = n_hard + n_easy
        # members of each set first (stable), fg additionally in random order
        fg_order = torch.argsort(torch.where(fg, u_perm, u_perm + 2.0), dim=1)            # random permutation of fg first
        hard_order = torch.argsort((~hard).to(torch.int8), dim=1, stable=True)
        easy_order = torch.argsort((~easy).to(torch.int8), dim=1, stable=True)
        # quotas
        fg_take = torch.where(n_bg > 0, torch.clamp(n_fg, max=fg_quota), torch.full_like(n_fg, P))
        fg_take = torch.where(n_fg > 0, fg_take, torch.zeros_like(fg_take))
        bg_take = P - fg_take
        hard_take = torch.where(n_easy > 0, torch.minimum((bg_take.float() * cfg.HARD_BG_RATIO).long(), n_hard), bg_take)
        hard_take = torch.where(n_hard > 0, hard_take, torch.zeros_like(hard_take))
        slot = torch.arange(P, device=dev).view(1, P)
        in_fg = slot < fg_take[:, None]
        in_hard = (~in_fg) & (slot < (fg_take + hard_take)[:, None])
        # fg slots: without replacement while n_fg >= fg_take (bg exists), with replacement when fg fills every slot
        fg_pos = torch.where((n_bg > 0)[:, None], slot.expand(B, P),
                             (u_slot * n_fg.clamp(min=1)[:, None].float()).floor().long())
        fg_pos = fg_pos.clamp(max=R - 1)
        fg_idx = torch.gather(fg_order, 1, torch.minimum(fg_pos, (n_fg.clamp(min=1) - 1)[:, None]))
        hard_pos = (u_slot * n_hard.clamp(min=1)[:, None].float()).floor().long()
        hard_idx = torch.gather(hard_order, 1, torch.minimum(hard_pos, (n_hard.clamp(min=1) - 1)[:, None]))
        easy_pos = (u_slot * n_easy.clamp(min=1)[:, None].float()).floor().long()
        easy_idx = torch.gather(easy_order, 1, torch.minimum(easy_pos, (n_easy.clamp(min=1) - 1)[:, None]))
        return torch.where(in_fg, fg_idx, torch.where(in_hard, hard_idx, easy_idx))
