"""RoIHeadTemplate (pcdet/models/roi_heads/roi_head_template.py:11-380): proposal layer (batched HIP NMS, no host
loop), canonical target transform, second-stage losses incl. the CRB branch (`reg_sample_targets`), box decoding.
tb_dict values are detached 0-dim tensors (no .item() synchronisation in the training step)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...ops.iou3d_nms import iou3d_nms_utils
from ...utils import box_coder_utils, common_utils, loss_utils
from .target_assigner.proposal_target_layer import ProposalTargetLayer


# second-stage losses + the canonical target transformation as HIP launches (crbhip.rcnn_loss: one launch each instead of ~270
# elementwise / reduction launches of a PV-RCNN step); CRB_RCNN_LOSS_FUSED=0 = the torch expressions below (A/B, and what the
# kernels are tested against)
FUSED_LOSS = __import__('os').environ.get('CRB_RCNN_LOSS_FUSED', '1') == '1'
# gathers of the proposal layer behind its NMS as one launch (csrc/proposal_layer.hip); CRB_PROPOSAL_FUSED=0 = torch expressions
FUSED_PROPOSAL = __import__('os').environ.get('CRB_PROPOSAL_FUSED', '1') == '1'


class RoIHeadTemplate(nn.Module):
    def __init__(self, num_class, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.box_coder = getattr(box_coder_utils, self.model_cfg.TARGET_CONFIG.BOX_CODER)(
            **self.model_cfg.TARGET_CONFIG.get('BOX_CODER_CONFIG', {}))
        self.proposal_target_layer = ProposalTargetLayer(roi_sampler_cfg=self.model_cfg.TARGET_CONFIG)
        self.build_losses(self.model_cfg.LOSS_CONFIG)
        self.forward_ret_dict = None

    def build_losses(self, losses_cfg):
        self.add_module('reg_loss_func',
                        loss_utils.WeightedSmoothL1Loss(code_weights=losses_cfg.LOSS_WEIGHTS['code_weights']))

    def make_fc_layers(self, input_channels, output_channels, fc_list):
        layers, pre = [], input_channels
        for k, c in enumerate(fc_list):
            layers += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if self.model_cfg.DP_RATIO >= 0 and k == 0:
                layers.append(nn.Dropout(self.model_cfg.DP_RATIO))
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    @torch.no_grad()
    def proposal_layer(self, batch_dict, nms_config):
        """batch_cls_preds (B,A,C), batch_box_preds (B,A,7) -> rois (B,POST,7) zero padded, roi_scores, roi_labels (1-based,
        padding = 1 as in the reference), full_cls_scores (B,POST,C)   (roi_head_template.py:45-108).
        All frames go through one top-k, one batched mask launch and one on-device greedy scan."""
        if batch_dict.get('rois', None) is not None:
            return batch_dict
        if nms_config.MULTI_CLASSES_NMS:
            raise NotImplementedError
        box_preds, cls_preds = batch_dict.get('batch_box_preds', None), batch_dict['batch_cls_preds']
        assert cls_preds.dim() == 3 and batch_dict.get('batch_index', None) is None
        B, A = cls_preds.shape[0], cls_preds.shape[1]
        post = nms_config.NMS_POST_MAXSIZE
        scores, labels = torch.max(cls_preds, dim=2)
        k = min(nms_config.NMS_PRE_MAXSIZE, A)
        top_scores, top_idx = torch.topk(scores, k=k, dim=1)                       # sorted descending
        if box_preds is None:           # the dense head left the decode to us: only the k anchors kept here become boxes
            top_boxes = batch_dict['batch_box_decoder'](top_idx)
        else:
            top_boxes = torch.gather(box_preds, 1, top_idx[..., None].expand(-1, -1, box_preds.shape[-1]))
        keep, _ = iou3d_nms_utils.nms_batched(top_boxes[..., 0:7].contiguous(), None, nms_config.NMS_THRESH, post,
                                              rotated=(nms_config.NMS_TYPE == 'nms_gpu'))
        if FUSED_PROPOSAL and cls_preds.is_cuda and top_idx.dtype == torch.int64:
            from crbhip import lib, check, ptr, cur_stream
            C, bc, dev = int(cls_preds.shape[-1]), int(top_boxes.shape[-1]), cls_preds.device
            rois = torch.empty((B, post, bc), dtype=torch.float32, device=dev)
            roi_scores = torch.empty((B, post), dtype=torch.float32, device=dev)
            roi_labels = torch.empty((B, post), dtype=torch.int64, device=dev)
            full = torch.empty((B, post, C), dtype=torch.float32, device=dev)
            check(lib.crb_proposal_finish(ptr(keep), ptr(top_idx.contiguous()), ptr(top_boxes.contiguous().float()),
                                          ptr(scores.contiguous().float()), ptr(labels.contiguous()), ptr(cls_preds.contiguous().float()),
                                          B, A, k, post, bc, C, ptr(rois), ptr(roi_scores), ptr(roi_labels), ptr(full), cur_stream(dev)),
                  'crb_proposal_finish')
            batch_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, full_cls_scores=full,
                              has_class_labels=True if C > 1 else False)
            batch_dict.pop('batch_index', None)
            return batch_dict
        valid = keep >= 0
        kept = keep.clamp(min=0).long()
        sel = torch.gather(top_idx, 1, kept)                                       # indices into the anchors
        vf = valid[..., None].to(top_boxes.dtype)
        rois = torch.gather(top_boxes, 1, kept[..., None].expand(-1, -1, top_boxes.shape[-1])) * vf
        roi_scores = torch.gather(scores, 1, sel) * valid.to(scores.dtype)
        roi_labels = torch.gather(labels, 1, sel) * valid.long()
        full = torch.gather(cls_preds, 1, sel[..., None].expand(-1, -1, cls_preds.shape[-1])) * vf
        batch_dict['rois'] = rois
        batch_dict['roi_scores'] = roi_scores
        batch_dict['roi_labels'] = roi_labels + 1
        batch_dict['full_cls_scores'] = full
        batch_dict['has_class_labels'] = True if cls_preds.shape[-1] > 1 else False
        batch_dict.pop('batch_index', None)
        return batch_dict

    def assign_targets(self, batch_dict):
        batch_size = batch_dict['batch_size']
        with torch.no_grad():
            # optional injected uniforms (u_perm (B,R), u_slot (B,P)): the batched CRB stage 2 draws them frame by frame
            targets_dict = self.proposal_target_layer.forward(batch_dict, batch_dict.get('roi_sampler_uniforms', None))
        rois = targets_dict['rois']
        gt = targets_dict['gt_of_rois']
        targets_dict['gt_of_rois_src'] = gt.clone().detach()
        if FUSED_LOSS and gt.is_cuda:
            from crbhip import rcnn_loss
            targets_dict['gt_of_rois'] = rcnn_loss.roi_canonical_targets(rois, gt)
            return targets_dict
        # canonical transformation: into the RoI frame (roi_head_template.py:118-138)
        roi_ry = rois[:, :, 6] % (2 * np.pi)
        center = gt[:, :, 0:3] - rois[:, :, 0:3]
        ry = gt[:, :, 6] - roi_ry
        local = torch.cat([center, gt[:, :, 3:6], ry[..., None], gt[:, :, 7:]], dim=-1)
        local = common_utils.rotate_points_along_z(points=local.view(-1, 1, local.shape[-1]),
                                                   angle=-roi_ry.view(-1)).view(batch_size, -1, local.shape[-1])
        heading = local[:, :, 6] % (2 * np.pi)
        opposite = (heading > np.pi * 0.5) & (heading < np.pi * 1.5)
        heading = torch.where(opposite, (heading + np.pi) % (2 * np.pi), heading)
        heading = torch.where(heading > np.pi, heading - np.pi * 2, heading)
        heading = torch.clamp(heading, min=-np.pi / 2, max=np.pi / 2)
        targets_dict['gt_of_rois'] = torch.cat([local[:, :, :6], heading[..., None], local[:, :, 7:]], dim=-1)
        return targets_dict

    def get_box_reg_layer_loss(self, forward_ret_dict, reduce=True):
        loss_cfgs = self.model_cfg.LOSS_CONFIG
        rcnn_reg = forward_ret_dict['rcnn_reg']
        if 'reg_sample_targets' in forward_ret_dict.keys():
            # CRB stage 2: regress towards the stage-1 hypothetical labels; returns the unreduced (1,M,7) loss tensor
            # exactly like the reference branch (roi_head_template.py:146-156)
            assert loss_cfgs.REG_LOSS == 'smooth-l1'
            loss = self.reg_loss_func(rcnn_reg.unsqueeze(0), forward_ret_dict['reg_sample_targets'].unsqueeze(0))
            return loss * loss_cfgs.LOSS_WEIGHTS['rcnn_reg_weight']
        if loss_cfgs.REG_LOSS != 'smooth-l1':
            raise NotImplementedError
        code_size = self.box_coder.code_size
        reg_valid_mask = forward_ret_dict['reg_valid_mask'].view(-1)
        gt_ct = forward_ret_dict['gt_of_rois'][..., 0:code_size]
        gt_src = forward_ret_dict['gt_of_rois_src'][..., 0:code_size].view(-1, code_size)
        roi_boxes3d = forward_ret_dict['rois']
        batch_size = gt_ct.shape[0]
        n = gt_ct.view(-1, code_size).shape[0]
        fg = (reg_valid_mask > 0)
        fgf = fg.float()
        fg_sum = fgf.sum()
        denom = torch.clamp(fg_sum, min=1.0)
        rois_anchor = roi_boxes3d.clone().detach().view(-1, code_size)
        rois_anchor = torch.cat([torch.zeros_like(rois_anchor[:, 0:3]), rois_anchor[:, 3:6],
                                 torch.zeros_like(rois_anchor[:, 6:7]), rois_anchor[:, 7:]], dim=-1)
        reg_targets = self.box_coder.encode_torch(gt_ct.view(n, code_size), rois_anchor)
        forward_ret_dict['rcnn_reg_gt'] = reg_targets
        loss = self.reg_loss_func(rcnn_reg.view(n, -1).unsqueeze(0), reg_targets.unsqueeze(0)).view(n, -1)
        if reduce:
            loss_reg = (loss * fgf.unsqueeze(-1)).sum() / denom
        else:
            loss_reg = ((loss * fgf.unsqueeze(-1)) / denom).view(batch_size, -1).sum(-1)
        loss_reg = loss_reg * loss_cfgs.LOSS_WEIGHTS['rcnn_reg_weight']
        tb = {'rcnn_loss_reg': (loss_reg if reduce else loss_reg[0]).detach()}
        if loss_cfgs.CORNER_LOSS_REGULARIZATION:
            # masked form of the fg-only corner loss: every RoI is decoded, background rows get weight 0
            rois_flat = roi_boxes3d.view(-1, code_size)
            anchors = torch.cat([torch.zeros_like(rois_flat[:, 0:3]), rois_flat[:, 3:]], dim=-1).detach()
            boxes = self.box_coder.decode_torch(rcnn_reg.view(n, code_size), anchors)
            boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(1), rois_flat[:, 6]).squeeze(1)
            boxes = torch.cat([boxes[:, 0:3] + rois_flat[:, 0:3], boxes[:, 3:]], dim=-1)
            corner = loss_utils.get_corner_loss_lidar(boxes[:, 0:7], gt_src[:, 0:7])
            corner = torch.where(fg, corner, torch.zeros_like(corner))
            if reduce:
                loss_corner = corner.sum() / denom
            else:
                per = fgf.view(batch_size, -1).sum(-1).clamp(min=1.0)
                loss_corner = corner.view(batch_size, -1).sum(-1) / per
            loss_corner = loss_corner * loss_cfgs.LOSS_WEIGHTS['rcnn_corner_weight']
            loss_reg = loss_reg + loss_corner
            tb['rcnn_loss_corner'] = (loss_corner if reduce else loss_corner[0]).detach()
        return loss_reg, tb

    def get_box_cls_layer_loss(self, forward_ret_dict, reduce=True):
        loss_cfgs = self.model_cfg.LOSS_CONFIG
        rcnn_cls = forward_ret_dict['rcnn_cls']
        labels = forward_ret_dict['rcnn_cls_labels']
        batch_size = labels.shape[0]
        labels = labels.view(-1)
        if loss_cfgs.CLS_LOSS == 'BinaryCrossEntropy':
            # (ignored RoIs carry the label -1 and weight 0: clamped for the call, whose device kernel asserts targets in [0, 1])
            batch_loss = F.binary_cross_entropy(torch.sigmoid(rcnn_cls.view(-1)), labels.float().clamp(min=0), reduction='none')
            valid = (labels >= 0).float()
            if reduce:
                loss = (batch_loss * valid).sum() / torch.clamp(valid.sum(), min=1.0)
            else:
                loss = ((batch_loss * valid) / torch.clamp(valid.sum(), min=1.0)).view(batch_size, -1).sum(-1)
        elif loss_cfgs.CLS_LOSS == 'CrossEntropy':
            batch_loss = F.cross_entropy(rcnn_cls, labels, reduction='none', ignore_index=-1)
            valid = (labels >= 0).float()
            loss = (batch_loss * valid).sum() / torch.clamp(valid.sum(), min=1.0)
        else:
            raise NotImplementedError
        loss = loss * loss_cfgs.LOSS_WEIGHTS['rcnn_cls_weight']
        return loss, {'rcnn_loss_cls': (loss if reduce else loss[0]).detach()}

    def _fused_loss_cfg(self, reduce):
        """the configuration crb_rcnn_loss implements (BinaryCrossEntropy, smooth-l1 (+ corner), code size 7, the whole-batch
        reduction, not the CRB branch) -> CrbRcnnLossCfg or None"""
        loss_cfgs, ret = self.model_cfg.LOSS_CONFIG, self.forward_ret_dict
        if not (FUSED_LOSS and reduce) or 'reg_sample_targets' in ret or not ret['rcnn_reg'].is_cuda or \
                loss_cfgs.CLS_LOSS != 'BinaryCrossEntropy' or loss_cfgs.REG_LOSS != 'smooth-l1' or self.box_coder.code_size != 7 or \
                getattr(self.box_coder, 'encode_angle_by_sincos', False) or ret['rcnn_reg'].shape[-1] != 7 or \
                ret['rcnn_cls'].numel() != ret['rcnn_reg'].shape[0] or self.reg_loss_func.code_weights is None:
            return None
        key = (float(self.reg_loss_func.beta), tuple(float(w) for w in loss_cfgs.LOSS_WEIGHTS['code_weights']),
               float(loss_cfgs.LOSS_WEIGHTS['rcnn_cls_weight']), float(loss_cfgs.LOSS_WEIGHTS['rcnn_reg_weight']),
               float(loss_cfgs.LOSS_WEIGHTS.get('rcnn_corner_weight', 0.0)), bool(loss_cfgs.CORNER_LOSS_REGULARIZATION))
        hit = self.__dict__.get('_crb_rcnn_cfg')
        if hit is None or hit[0] != key:
            from crbhip import rcnn_loss
            hit = self.__dict__['_crb_rcnn_cfg'] = (key, rcnn_loss.make_cfg(key[1], key[2], key[3], key[4], key[5], beta=key[0]))
        return hit[1]

    def get_loss(self, tb_dict=None, reduce=True):
        tb_dict = {} if tb_dict is None else tb_dict
        cfg = self._fused_loss_cfg(reduce)
        if cfg is not None:
            from crbhip import rcnn_loss
            ret = self.forward_ret_dict
            total, parts, ret['rcnn_reg_gt'] = rcnn_loss.rcnn_loss(ret['rcnn_cls'], ret['rcnn_reg'], ret['rcnn_cls_labels'],
                                                                   ret['reg_valid_mask'], ret['rois'], ret['gt_of_rois'],
                                                                   ret['gt_of_rois_src'], cfg)
            tb_dict.update({'rcnn_loss_cls': parts[0], 'rcnn_loss_reg': parts[1], 'rcnn_loss': parts[3]})
            if cfg.corner:
                tb_dict['rcnn_loss_corner'] = parts[2]
            return total, tb_dict
        loss_cls, tb1 = self.get_box_cls_layer_loss(self.forward_ret_dict, reduce=reduce)
        loss_reg, tb2 = self.get_box_reg_layer_loss(self.forward_ret_dict, reduce=reduce)
        tb_dict.update(tb1)
        tb_dict.update(tb2)
        rcnn_loss = loss_cls + loss_reg
        tb_dict['rcnn_loss'] = (rcnn_loss if reduce else rcnn_loss[0]).detach()
        return rcnn_loss, tb_dict

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        """rois (B,N,7), cls (BN,C), box (BN,code) -> (B,N,C), (B,N,code) in LiDAR coordinates"""
        code_size = self.box_coder.code_size
        batch_cls_preds = cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        if FUSED_PROPOSAL and box_preds.is_cuda and code_size == 7 and not torch.is_grad_enabled() and \
                not getattr(self.box_coder, 'encode_angle_by_sincos', False) and box_preds.shape[-1] == 7:
            from crbhip import lib, check, ptr, cur_stream
            r = rois.contiguous().float()
            n = box_preds.numel() // 7
            out = torch.empty((batch_size, n // batch_size, 7), dtype=torch.float32, device=box_preds.device)
            check(lib.crb_rcnn_decode_boxes(ptr(r), int(r.shape[-1]), ptr(box_preds.contiguous().float()), n, ptr(out),
                                            cur_stream(box_preds.device)), 'crb_rcnn_decode_boxes')
            return batch_cls_preds, out
        local_rois = torch.cat([torch.zeros_like(rois[:, :, 0:3]), rois[:, :, 3:]], dim=-1).detach()
        boxes = self.box_coder.decode_torch(box_preds.view(batch_size, -1, code_size), local_rois).view(-1, code_size)
        boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(1), rois[:, :, 6].reshape(-1)).squeeze(1)
        boxes = torch.cat([boxes[:, 0:3] + rois[:, :, 0:3].reshape(-1, 3), boxes[:, 3:]], dim=-1)
        return batch_cls_preds, boxes.view(batch_size, -1, code_size)
