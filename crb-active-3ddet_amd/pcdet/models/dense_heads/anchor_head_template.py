"""AnchorHeadTemplate (pcdet/models/dense_heads/anchor_head_template.py:11-285): anchors, target assignment, RPN losses
and box decoding. Differences from the reference are host-side only: anchors are registered as non-persistent buffers
(no .cuda() in the constructor, anchor_head_template.py:31) and the loss path never calls .item() — tb_dict values are
detached 0-dim tensors (float()-convertible) so a training step stays free of host synchronisation."""
import numpy as np
import torch
import torch.nn as nn

from ...utils import box_coder_utils, common_utils, loss_utils
from .target_assigner.anchor_generator import AnchorGenerator
from .target_assigner.axis_aligned_target_assigner import AxisAlignedTargetAssigner


FUSED_LOSS = True     # RPN losses through crb_rpn_loss_forward / _backward for device tensors; False = the torch restatement


# decode of the proposal layer's top-k anchors as one launch (CRB_PROPOSAL_FUSED=0: the torch expressions, A/B and test reference)
FUSED_DECODE = __import__('os').environ.get('CRB_PROPOSAL_FUSED', '1') == '1'


class AnchorHeadTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, class_names, grid_size, point_cloud_range, predict_boxes_when_training):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.class_names = class_names
        self.predict_boxes_when_training = predict_boxes_when_training
        self.use_multihead = self.model_cfg.get('USE_MULTIHEAD', False)

        anchor_target_cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        self.box_coder = getattr(box_coder_utils, anchor_target_cfg.BOX_CODER)(
            num_dir_bins=anchor_target_cfg.get('NUM_DIR_BINS', 6), **anchor_target_cfg.get('BOX_CODER_CONFIG', {}))

        anchors, self.num_anchors_per_location = self.generate_anchors(
            self.model_cfg.ANCHOR_GENERATOR_CONFIG, grid_size=grid_size, point_cloud_range=point_cloud_range,
            anchor_ndim=self.box_coder.code_size)
        self._n_anchor_sets = len(anchors)
        for i, a in enumerate(anchors):
            self.register_buffer('_anchors_%d' % i, a, persistent=False)
        self.target_assigner = self.get_target_assigner(anchor_target_cfg)
        self.forward_ret_dict = {}
        self.build_losses(self.model_cfg.LOSS_CONFIG)

    @property
    def anchors(self):
        return [getattr(self, '_anchors_%d' % i) for i in range(self._n_anchor_sets)]

    @staticmethod
    def generate_anchors(anchor_generator_cfg, grid_size, point_cloud_range, anchor_ndim=7):
        gen = AnchorGenerator(anchor_range=point_cloud_range, anchor_generator_config=anchor_generator_cfg)
        gs = np.asarray(grid_size)
        feature_map_size = [gs[:2] // c['feature_map_stride'] for c in anchor_generator_cfg]
        anchors_list, per_loc = gen.generate_anchors(feature_map_size)
        if anchor_ndim != 7:
            anchors_list = [torch.cat((a, a.new_zeros([*a.shape[:-1], anchor_ndim - 7])), dim=-1)
                            for a in anchors_list]
        return anchors_list, per_loc

    def get_target_assigner(self, anchor_target_cfg):
        if anchor_target_cfg.NAME != 'AxisAlignedTargetAssigner':
            raise NotImplementedError(anchor_target_cfg.NAME)
        return AxisAlignedTargetAssigner(model_cfg=self.model_cfg, class_names=self.class_names,
                                         box_coder=self.box_coder, match_height=anchor_target_cfg.MATCH_HEIGHT)

    def build_losses(self, losses_cfg):
        self.add_module('cls_loss_func', loss_utils.SigmoidFocalClassificationLoss(alpha=0.25, gamma=2.0))
        reg_name = losses_cfg.get('REG_LOSS_TYPE', None) or 'WeightedSmoothL1Loss'
        self.add_module('reg_loss_func',
                        getattr(loss_utils, reg_name)(code_weights=losses_cfg.LOSS_WEIGHTS['code_weights']))
        self.add_module('dir_loss_func', loss_utils.WeightedCrossEntropyLoss())

    def assign_targets(self, gt_boxes):
        return self.target_assigner.assign_targets(self.anchors, gt_boxes)

    def _flat_anchors(self):
        """the per-class anchor maps side by side (anchor_head_template.py:245); constants of the head: built once per device"""
        hit = self.__dict__.get('_crb_flat_anchors')
        if hit is None or hit[0] is not self.anchors or hit[1].device != self.anchors[0].device:
            hit = self.__dict__['_crb_flat_anchors'] = (self.anchors, torch.cat(self.anchors, dim=-3))
        return hit[1]

    def get_cls_layer_loss(self, new_data=None, reduce=True):
        src = self.forward_ret_dict if new_data is None else new_data
        cls_preds, box_cls_labels = src['cls_preds'], src['box_cls_labels']
        B = int(cls_preds.shape[0])
        cared = box_cls_labels >= 0
        positives = box_cls_labels > 0
        negatives = box_cls_labels == 0
        cls_weights = (negatives * 1.0 + 1.0 * positives).float()
        pos_normalizer = positives.sum(1, keepdim=True).float()
        cls_weights = cls_weights / torch.clamp(pos_normalizer, min=1.0)
        if self.num_class == 1:
            box_cls_labels = torch.where(positives, torch.ones_like(box_cls_labels), box_cls_labels)
        cls_targets = (box_cls_labels * cared.type_as(box_cls_labels)).long()
        one_hot = torch.zeros(*cls_targets.shape, self.num_class + 1, dtype=cls_preds.dtype, device=cls_preds.device)
        one_hot.scatter_(-1, cls_targets.unsqueeze(-1), 1.0)
        loss_src = self.cls_loss_func(cls_preds.view(B, -1, self.num_class), one_hot[..., 1:], weights=cls_weights)
        cls_loss = loss_src.sum() / B if reduce else loss_src.sum(-1).sum(-1)
        cls_loss = cls_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['cls_weight']
        return cls_loss, {'rpn_loss_cls': (cls_loss if reduce else cls_loss[0]).detach()}

    @staticmethod
    def add_sin_difference(boxes1, boxes2, dim=6):
        assert dim != -1
        a, b = boxes1[..., dim:dim + 1], boxes2[..., dim:dim + 1]
        enc1 = torch.sin(a) * torch.cos(b)
        enc2 = torch.cos(a) * torch.sin(b)
        boxes1 = torch.cat([boxes1[..., :dim], enc1, boxes1[..., dim + 1:]], dim=-1)
        boxes2 = torch.cat([boxes2[..., :dim], enc2, boxes2[..., dim + 1:]], dim=-1)
        return boxes1, boxes2

    @staticmethod
    def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0, num_bins=2):
        B = reg_targets.shape[0]
        anchors = anchors.view(B, -1, anchors.shape[-1])
        rot_gt = reg_targets[..., 6] + anchors[..., 6]
        offset_rot = common_utils.limit_period(rot_gt - dir_offset, 0, 2 * np.pi)
        dir_cls = torch.clamp(torch.floor(offset_rot / (2 * np.pi / num_bins)).long(), min=0, max=num_bins - 1)
        if one_hot:
            t = torch.zeros(*dir_cls.shape, num_bins, dtype=anchors.dtype, device=dir_cls.device)
            t.scatter_(-1, dir_cls.unsqueeze(-1), 1.0)
            return t
        return dir_cls

    def get_box_reg_layer_loss(self, reduce=True):
        d = self.forward_ret_dict
        box_preds, dir_preds = d['box_preds'], d.get('dir_cls_preds', None)
        reg_targets, labels = d['box_reg_targets'], d['box_cls_labels']
        B = int(box_preds.shape[0])
        positives = labels > 0
        reg_weights = positives.float()
        reg_weights = reg_weights / torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
        anchors = self._flat_anchors()
        anchors = anchors.view(1, -1, anchors.shape[-1]).expand(B, -1, -1)
        box_preds = box_preds.view(B, -1, box_preds.shape[-1] // self.num_anchors_per_location)
        p_sin, t_sin = self.add_sin_difference(box_preds, reg_targets)
        loc_src = self.reg_loss_func(p_sin, t_sin, weights=reg_weights)
        loc_loss = loc_src.sum() / B if reduce else loc_src.sum(-1).sum(-1)
        loc_loss = loc_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['loc_weight']
        box_loss = loc_loss
        tb = {'rpn_loss_loc': (loc_loss if reduce else loc_loss[0]).detach()}
        if dir_preds is not None:
            dir_targets = self.get_direction_target(anchors, reg_targets, dir_offset=self.model_cfg.DIR_OFFSET,
                                                    num_bins=self.model_cfg.NUM_DIR_BINS)
            dir_logits = dir_preds.view(B, -1, self.model_cfg.NUM_DIR_BINS)
            w = positives.type_as(dir_logits)
            w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.dir_loss_func(dir_logits, dir_targets, weights=w)
            dir_loss = dir_loss.sum() / B if reduce else dir_loss.sum(-1).sum(-1)
            dir_loss = dir_loss * self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS['dir_weight']
            box_loss = box_loss + dir_loss
            # reduce=False: the reference sums the (B,A) direction loss over BOTH axes (anchor_head_template.py:219-225), i.e.
            # every frame's box loss carries the direction loss of the whole batch; kept as it is
            tb['rpn_loss_dir'] = dir_loss.detach()
        return box_loss, tb

    def _fused_loss_cfg(self):
        """CrbRpnLossCfg for the HIP loss kernels, or None when this head's loss configuration is not the one they implement
        (focal classification + WeightedSmoothL1Loss with 7 code weights + optional direction cross entropy)"""
        lw = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        cw = getattr(self.reg_loss_func, 'code_weights', None)
        # the struct is cached, keyed on every value it snapshots (a load_state_dict, a config edit between active-learning rounds
        # or a test that flips a weight must not be ignored by the fused path: ADVICE r03)
        key = (self.num_class, self.model_cfg.get('NUM_DIR_BINS', 2), lw['cls_weight'], lw['loc_weight'], lw.get('dir_weight', 0.0),
               self.model_cfg.get('DIR_OFFSET', 0.0), getattr(self.cls_loss_func, 'alpha', None), getattr(self.cls_loss_func, 'gamma', None),
               getattr(self.reg_loss_func, 'beta', None), None if cw is None else (cw.data_ptr(), cw._version),
               type(self.reg_loss_func), type(self.cls_loss_func))
        if getattr(self, '_rpn_loss_key', None) != key:
            self._rpn_loss_key, self._rpn_loss_cfg = key, None
        if getattr(self, '_rpn_loss_cfg', None) is None:
            from crbhip import rpn_loss as _rl
            ok = (type(self.reg_loss_func) is loss_utils.WeightedSmoothL1Loss and self.box_coder.code_size == 7
                  and type(self.cls_loss_func) is loss_utils.SigmoidFocalClassificationLoss and self.num_class <= 8
                  and self.reg_loss_func.code_weights is not None and self.model_cfg.get('NUM_DIR_BINS', 2) <= 8)
            if not ok:
                self._rpn_loss_cfg = False
            else:
                self._rpn_loss_cfg = _rl.make_cfg(
                    self.num_class, self.model_cfg.get('NUM_DIR_BINS', 2), self.reg_loss_func.code_weights.tolist(),
                    lw['cls_weight'], lw['loc_weight'], lw.get('dir_weight', 0.0), self.model_cfg.get('DIR_OFFSET', 0.0),
                    alpha=self.cls_loss_func.alpha, gamma=self.cls_loss_func.gamma, beta=self.reg_loss_func.beta)
        return self._rpn_loss_cfg or None

    def get_loss_fused(self, cfg, reduce=True):
        """get_cls_layer_loss + get_box_reg_layer_loss as one forward and one backward HIP launch (csrc/rpn_loss.hip):
        same three losses (sum of the same per-anchor terms, normalised by the frame's positives), same tb_dict"""
        from crbhip import rpn_loss as _rl
        d = self.forward_ret_dict
        B = int(d['cls_preds'].shape[0])
        anchors = self._flat_anchors().reshape(-1, 7)
        per_frame = _rl.rpn_loss(d['cls_preds'], d['box_preds'], d.get('dir_cls_preds', None), d['box_cls_labels'],
                                 d['box_reg_targets'], anchors, cfg)                       # (B,3)
        has_dir = d.get('dir_cls_preds', None) is not None
        if reduce:
            parts = per_frame.sum(0) / B                                # (3) = the reference's three scalars
            det = parts.detach()
            tb = {'rpn_loss_cls': det[0], 'rpn_loss_loc': det[1]}
            rpn_loss = parts.sum() if has_dir else parts[:2].sum()      # one reduction node in the graph, no per-term selects
            if has_dir:
                tb['rpn_loss_dir'] = det[2]
            tb['rpn_loss'] = rpn_loss.detach()
            return rpn_loss, tb
        cls_loss, loc_loss = per_frame[:, 0], per_frame[:, 1]
        tb = {'rpn_loss_cls': cls_loss[0].detach(), 'rpn_loss_loc': loc_loss[0].detach()}
        rpn_loss = cls_loss + loc_loss
        if has_dir:
            dir_loss = per_frame[:, 2].sum()       # reduce=False: summed over the frames as well, as the reference does
            tb['rpn_loss_dir'] = dir_loss.detach()
            rpn_loss = rpn_loss + dir_loss
        tb['rpn_loss'] = rpn_loss[0].detach()
        return rpn_loss, tb

    def get_loss(self, reduce=True):
        if FUSED_LOSS and self.forward_ret_dict['cls_preds'].is_cuda:
            cfg = self._fused_loss_cfg()
            if cfg is not None:
                return self.get_loss_fused(cfg, reduce=reduce)
        cls_loss, tb = self.get_cls_layer_loss(reduce=reduce)
        box_loss, tb_box = self.get_box_reg_layer_loss(reduce=reduce)
        tb.update(tb_box)
        rpn_loss = cls_loss + box_loss
        tb['rpn_loss'] = (rpn_loss if reduce else rpn_loss[0]).detach()
        return rpn_loss, tb

    def generate_predicted_boxes(self, batch_size, cls_preds, box_preds, dir_cls_preds=None, anchor_idx=None):
        """cls (B,H,W,C1), box (B,H,W,C2), dir (B,H,W,C3) -> (B,A,num_class), (B,A,7+C)  (anchor_head_template.py:238-285).
        anchor_idx (B,k) long: decode ONLY those anchors -> (B,A,num_class), (B,k,7+C) — the rows the full decode would hold at
        these indices (the decode is elementwise per anchor). A two-stage detector reads the RPN boxes through the proposal
        layer's top-k alone: 9,000 (training) / 1,024 (test) of the 211,200 anchors per frame."""
        anchors = self._flat_anchors()
        A = anchors.view(-1, anchors.shape[-1]).shape[0]
        batch_cls_preds = cls_preds.view(batch_size, A, -1).float()
        raw = box_preds.view(batch_size, A, -1)
        dirs = dir_cls_preds.view(batch_size, A, -1) if dir_cls_preds is not None else None
        if anchor_idx is not None and FUSED_DECODE and raw.is_cuda and raw.shape[-1] == 7 and anchors.shape[-1] == 7 and \
                type(self.box_coder).__name__ == 'ResidualCoder' and not getattr(self.box_coder, 'encode_angle_by_sincos', False):
            # the top-k anchors of the proposal layer: gather + decode + direction bins as one launch (csrc/proposal_layer.hip)
            from crbhip import lib, check, ptr, cur_stream
            k = int(anchor_idx.shape[1])
            out = torch.empty((batch_size, k, 7), dtype=torch.float32, device=raw.device)
            check(lib.crb_decode_selected_anchors(ptr(raw.contiguous().float()), ptr(None if dirs is None else dirs.contiguous().float()),
                                                  ptr(anchors.view(-1, 7).contiguous().float()), ptr(anchor_idx.contiguous()), batch_size,
                                                  A, k, 0 if dirs is None else int(dirs.shape[-1]), float(self.model_cfg.get('DIR_OFFSET', 0.0)),
                                                  float(self.model_cfg.get('DIR_LIMIT_OFFSET', 0.0)), ptr(out), cur_stream(raw.device)),
                  'crb_decode_selected_anchors')
            return batch_cls_preds, out
        if anchor_idx is None:
            batch_anchors = anchors.view(1, -1, anchors.shape[-1]).expand(batch_size, -1, -1)
        else:
            batch_anchors = anchors.view(-1, anchors.shape[-1])[anchor_idx]                               # (B,k,7)
            raw = torch.gather(raw, 1, anchor_idx[..., None].expand(-1, -1, raw.shape[-1]))
            if dirs is not None:
                dirs = torch.gather(dirs, 1, anchor_idx[..., None].expand(-1, -1, dirs.shape[-1]))
        batch_box_preds = self.box_coder.decode_torch(raw, batch_anchors)
        if dirs is not None:
            dir_offset, dir_limit_offset = self.model_cfg.DIR_OFFSET, self.model_cfg.DIR_LIMIT_OFFSET
            dir_labels = torch.max(dirs, dim=-1)[1]
            period = 2 * np.pi / self.model_cfg.NUM_DIR_BINS
            dir_rot = common_utils.limit_period(batch_box_preds[..., 6] - dir_offset, dir_limit_offset, period)
            rot = dir_rot + dir_offset + period * dir_labels.to(batch_box_preds.dtype)
            batch_box_preds = torch.cat([batch_box_preds[..., :6], rot.unsqueeze(-1), batch_box_preds[..., 7:]], dim=-1)
        return batch_cls_preds, batch_box_preds

    def forward(self, **kwargs):
        raise NotImplementedError
