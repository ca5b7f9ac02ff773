"""PartA2FCHead (pcdet/models/roi_heads/partA2_head.py:10-224): RoI-aware pooling of the point head's part offsets (avg)
and point features (max) into an out^3 grid per RoI, two SubM conv stacks over the occupied cells of all B*N grids as ONE
sparse tensor (batch index = RoI), dense flatten, shared FC, class / box branches.

Everything numeric runs on the HIP entry points the other heads already use: crb_roiaware_pool3d_forward / _backward
(RoIAwarePool3d), the subm rulebook + gather-GEMM of the spconv mirror (4->64, 64->64, C_in->64), crb_sparse_to_dense.
The detector around it in the reference (UNetV2 + PointIntraPartOffsetHead, part_a2_net.py) is out of scope (SURVEY §2.1
row 8); the head consumes `point_part_offset` / `point_cls_scores` / `point_features` from whichever point head fills them."""
import torch
import torch.nn as nn

from ...ops.roiaware_pool3d import roiaware_pool3d_utils
from ...utils.spconv_utils import spconv
from .roi_head_template import RoIHeadTemplate


class PartA2FCHead(RoIHeadTemplate):
    def __init__(self, input_channels, model_cfg, num_class=1, **kwargs):
        super().__init__(num_class=num_class, model_cfg=model_cfg)
        self.model_cfg = model_cfg
        pool_cfg = self.model_cfg.ROI_AWARE_POOL
        c0 = pool_cfg.NUM_FEATURES // 2
        block = self.post_act_block
        self.conv_part = spconv.SparseSequential(block(4, 64, 3, padding=1, indice_key='rcnn_subm1'),
                                                 block(64, c0, 3, padding=1, indice_key='rcnn_subm1_1'))
        self.conv_rpn = spconv.SparseSequential(block(input_channels, 64, 3, padding=1, indice_key='rcnn_subm2'),
                                                block(64, c0, 3, padding=1, indice_key='rcnn_subm1_2'))
        pre = pool_cfg.NUM_FEATURES * pool_cfg.POOL_SIZE ** 3
        shared = []
        n_fc = len(self.model_cfg.SHARED_FC)
        for k, c in enumerate(self.model_cfg.SHARED_FC):
            shared += [nn.Conv1d(pre, c, kernel_size=1, bias=False), nn.BatchNorm1d(c), nn.ReLU()]
            pre = c
            if k != n_fc - 1 and self.model_cfg.DP_RATIO > 0:
                shared.append(nn.Dropout(self.model_cfg.DP_RATIO))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.cls_layers = self.make_fc_layers(input_channels=pre, output_channels=self.num_class,
                                              fc_list=self.model_cfg.CLS_FC)
        self.reg_layers = self.make_fc_layers(input_channels=pre, output_channels=self.box_coder.code_size * self.num_class,
                                              fc_list=self.model_cfg.REG_FC)
        self.roiaware_pool3d_layer = roiaware_pool3d_utils.RoIAwarePool3d(
            out_size=pool_cfg.POOL_SIZE, max_pts_each_voxel=pool_cfg.MAX_POINTS_PER_VOXEL)
        self.init_weights(weight_init='xavier')

    def init_weights(self, weight_init='xavier'):
        init = {'kaiming': nn.init.kaiming_normal_, 'xavier': nn.init.xavier_normal_, 'normal': nn.init.normal_}[weight_init]
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                if weight_init == 'normal':
                    init(m.weight, mean=0, std=0.001)
                else:
                    init(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_layers[-1].weight, mean=0, std=0.001)

    @staticmethod
    def post_act_block(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0, conv_type='subm'):
        if conv_type == 'subm':
            conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
        elif conv_type == 'spconv':
            conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                       indice_key=indice_key)
        elif conv_type == 'inverseconv':
            conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
        else:
            raise NotImplementedError(conv_type)
        return spconv.SparseSequential(conv, nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01), nn.ReLU())

    def roiaware_pool(self, batch_dict):
        """-> pooled part features (B*N, o, o, o, 4) [avg of (part offset | xyz, score)], pooled point features
        (B*N, o, o, o, C) [max]; points below SEG_MASK_SCORE_THRESH contribute zero part offsets"""
        batch_size = batch_dict['batch_size']
        batch_idx = batch_dict['point_coords'][:, 0]
        xyz = batch_dict['point_coords'][:, 1:4]
        score = batch_dict['point_cls_scores'].view(-1, 1).detach()
        lead = xyz if self.model_cfg.get('DISABLE_PART', False) else batch_dict['point_part_offset']
        lead = torch.where(score < self.model_cfg.SEG_MASK_SCORE_THRESH, torch.zeros_like(lead), lead)
        part = torch.cat((lead, score), dim=1)
        feats = batch_dict['point_features']
        rois = batch_dict['rois']
        pooled_part, pooled_rpn = [], []
        for b in range(batch_size):
            m = batch_idx == b
            cur_xyz, cur_roi = xyz[m].contiguous(), rois[b][:, 0:7].contiguous()
            pooled_part.append(self.roiaware_pool3d_layer(cur_roi, cur_xyz, part[m].contiguous(), pool_method='avg'))
            pooled_rpn.append(self.roiaware_pool3d_layer(cur_roi, cur_xyz, feats[m].contiguous(), pool_method='max'))
        return torch.cat(pooled_part, dim=0), torch.cat(pooled_rpn, dim=0)

    @staticmethod
    def fake_sparse_idx(sparse_idx, batch_size_rcnn):
        """fewer than 3 occupied cells in the whole batch: cell (0,0,0) of every RoI stands in (BatchNorm needs values)"""
        z = sparse_idx.new_zeros((batch_size_rcnn, 3))
        return torch.cat((torch.arange(batch_size_rcnn, device=z.device).type_as(z).view(-1, 1), z), dim=1)

    def forward(self, batch_dict):
        targets_dict = self.proposal_layer(batch_dict,
                                           nms_config=self.model_cfg.NMS_CONFIG['TRAIN' if self.training else 'TEST'])
        if self.training:
            targets_dict = self.assign_targets(batch_dict)
            batch_dict['rois'] = targets_dict['rois']
            batch_dict['roi_labels'] = targets_dict['roi_labels']

        pooled_part, pooled_rpn = self.roiaware_pool(batch_dict)
        n_rcnn = pooled_part.shape[0]
        sparse_shape = [int(v) for v in pooled_part.shape[1:4]]
        sparse_idx = pooled_part.sum(dim=-1).nonzero()                  # (cells, 4): roi, x, y, z — lexicographic
        if sparse_idx.shape[0] < 3:
            sparse_idx = self.fake_sparse_idx(sparse_idx, n_rcnn)
            if self.training:                                            # nothing to learn from such a batch
                targets_dict['rcnn_cls_labels'].fill_(-1)
                targets_dict['reg_valid_mask'].fill_(-1)
        r, x, y, z = sparse_idx.unbind(1)
        coords = sparse_idx.int().contiguous()
        part = spconv.SparseConvTensor(pooled_part[r, x, y, z], coords, sparse_shape, n_rcnn)
        rpn = spconv.SparseConvTensor(pooled_rpn[r, x, y, z], coords, sparse_shape, n_rcnn)
        x_part = self.conv_part(part)
        x_rpn = self.conv_rpn(rpn)
        merged = torch.cat((x_rpn.features, x_part.features), dim=1)
        shared = spconv.SparseConvTensor(merged, coords, sparse_shape, n_rcnn).dense().view(n_rcnn, -1, 1)
        shared = self.shared_fc_layer(shared)
        rcnn_cls = self.cls_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)
        rcnn_reg = self.reg_layers(shared).transpose(1, 2).contiguous().squeeze(dim=1)

        if not self.training:
            cls_preds, box_preds = self.generate_predicted_boxes(batch_size=batch_dict['batch_size'], rois=batch_dict['rois'],
                                                                 cls_preds=rcnn_cls, box_preds=rcnn_reg)
            batch_dict['batch_cls_preds'] = cls_preds
            batch_dict['batch_box_preds'] = box_preds
            batch_dict['cls_preds_normalized'] = False
        else:
            targets_dict['rcnn_cls'] = rcnn_cls
            targets_dict['rcnn_reg'] = rcnn_reg
            self.forward_ret_dict = targets_dict
        return batch_dict
