"""Detector3DTemplate (pcdet/models/detectors/detector3d_template.py:14-500): module builder, CRB-patched
post_processing (active-learning record per frame) and checkpoint loading.

post_processing keeps the reference's per-frame record contract (the 15 keys of detector3d_template.py:390-406) but
computes the per-frame statistics with batched device ops: one points-in-boxes launch per box set instead of a Python
loop over classes / boxes with `(idx == i).sum()` host round trips (detector3d_template.py:249-261,381-387)."""
import os

import torch
import torch.nn as nn

from ...utils.spconv_utils import find_all_spconv_keys
from .. import backbones_2d, backbones_3d, dense_heads, roi_heads
from ..backbones_2d import map_to_bev
from ..backbones_3d import pfe, vfe
from ..model_utils import model_nms_utils


LAZY_BOX_DECODE = True


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.dataset = dataset
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d', 'dense_head',
                                'point_head', 'roi_head']

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def run_modules(self, batch_dict, before_pfe=None):
        """the detector's module chain on one batch (what every reference detector's forward() starts with); a PFE that can
        sample its keypoints ahead of time is told to start as soon as the points are known.
        before_pfe: optional callable run once between the dense half and the PFE (end of the chain without a PFE) - the place
        where a caller that pipelines batches enqueues the NEXT batch's prologue (prefetch_sparse): its farthest-point sampling then
        runs beside this batch's set-abstraction / RoI-head kernels instead of beside the persistent convolution kernels."""
        pfe_mod = getattr(self, 'pfe', None)
        if pfe_mod is not None and hasattr(pfe_mod, 'prefetch_keypoints') and '_keypoints_prefetched' not in batch_dict:
            pfe_mod.prefetch_keypoints(batch_dict)             # FPS on a side stream, joined inside the PFE
        if getattr(getattr(self, 'backbone_3d', None), 'ACCEPTS_LAZY_VOXELS', False) and 'voxels' not in batch_dict:
            batch_dict['_lazy_voxel_count'] = True             # one read-back for the voxel count + the table plan (mean_vfe.py)
        vfe_done = batch_dict.pop('_vfe_done', False)           # prefetch_sparse() ran the voxel generator for this batch
        for stage in self.scheduled_modules():
            if vfe_done and stage is getattr(self, 'vfe', None):
                continue
            if before_pfe is not None and stage is pfe_mod:
                before_pfe()
                before_pfe = None
            batch_dict = stage(batch_dict)
        if before_pfe is not None:
            before_pfe()
        return batch_dict

    def prefetch_sparse(self, batch_dict):
        """Software pipelining of the loader side of the step (the reference voxelizes in its DataLoader workers, ahead of the GPU
        step: pcdet/datasets/processor/data_processor.py:44-60): call with the NEXT batch right after this batch's forward pass has
        been enqueued (before backward()). The voxel generator and the marking half of the 3-D backbone's table plan run on the
        current stream without any host synchronisation, their counts travel to pinned memory; forward(batch_dict) of the next step
        picks them up (same dict object). Without it every forward pass reads the counts back in the middle of its sparse phase: the
        host waits for the device to drain and the device then waits for the host to refill the launch queue. Results are identical
        (same kernels on the same inputs). No-op for batches that carry loader-side voxels."""
        bb, vfe = getattr(self, 'backbone_3d', None), getattr(self, 'vfe', None)
        if bb is None or vfe is None or not getattr(bb, 'ACCEPTS_LAZY_VOXELS', False) or not hasattr(bb, 'prefetch') or \
                'voxels' in batch_dict or 'voxel_coords' in batch_dict or '_vfe_done' in batch_dict:
            return batch_dict
        import torch
        with torch.no_grad():
            batch_dict['_lazy_voxel_count'] = True
            vfe(batch_dict)
            if batch_dict.get('voxel_count_dev', None) is None:   # the generator answered with cut tensors: nothing to pipeline
                batch_dict['_vfe_done'] = True
                return batch_dict
            bb.prefetch(batch_dict)
            # the keypoint sampling of that batch (5 ms of sequential rounds, one workgroup per frame on a side stream) starts here
            # too: behind this batch's forward pass it runs beside the RoI-head / set-abstraction part of the backward pass. Started
            # with its own forward pass it ran beside the first persistent one-workgroup-per-CU convolution launches, and the one
            # that overlapped it took 2.7 ms instead of 0.73 (profiles/r05_pvrcnn_..._v3.csv launch list)
            pfe_mod = getattr(self, 'pfe', None)
            if pfe_mod is not None and hasattr(pfe_mod, 'prefetch_keypoints') and '_keypoints_prefetched' not in batch_dict:
                pfe_mod.prefetch_keypoints(batch_dict)
        batch_dict['_vfe_done'] = True
        return batch_dict

    DENSE_BEFORE_PFE = True

    def scheduled_modules(self):
        """module_list in execution order. The reference runs PFE -> BACKBONE_2D -> DENSE_HEAD (pv_rcnn.yaml); the PFE reads
        the BEV INPUT map and the 3D feature levels, the 2D backbone and the dense head read neither of its outputs, so the
        two groups commute. Running the dense half first puts 15-20 ms of MIOpen work between the launch of the keypoint
        sampling (5.4 ms of strictly sequential FPS rounds on a side stream, one workgroup per frame) and the first kernel
        that needs the keypoints; in the reference order the PFE waits for it."""
        mods = list(self.module_list)
        pfe = getattr(self, 'pfe', None)
        if not (self.DENSE_BEFORE_PFE and pfe is not None and hasattr(pfe, 'prefetch_keypoints') and pfe in mods):
            return mods
        i = mods.index(pfe)
        later = [m for m in mods[i + 1:] if m is getattr(self, 'backbone_2d', None) or m is getattr(self, 'dense_head', None)]
        if not later or any(m is not a for m, a in zip(mods[i + 1:], later)):      # only a contiguous [2D, head] run right after
            return mods
        return mods[:i] + later + [pfe] + mods[i + 1 + len(later):]

    @staticmethod
    def _batch_key(batch_dict):
        fid = batch_dict.get('frame_id', None)
        return None if fid is None else tuple(str(f) for f in fid)

    def _auto_prefetch(self, batch_dict):
        """the two halves of the look-ahead protocol (pcdet.datasets.LookaheadLoader): (1) this batch was prefetched during the last
        forward pass -> its device tensors and prologue state replace what the caller uploaded again (the batch is recognised by
        its frame ids, not by object identity: DistributedDataParallel hands forward() a copy of the dict); (2) the batch names
        its successor -> returns the callable that uploads it and enqueues its sparse prologue, run between this batch's dense
        half and its PFE. Batches without `_crb_next` / `frame_id`: nothing happens."""
        nxt = batch_dict.pop('_crb_next', None)
        held = self.__dict__.pop('_crb_prefetched', None)
        if held is not None and held[0] is not None and held[0] == self._batch_key(batch_dict) and '_vfe_done' not in batch_dict:
            batch_dict.update(held[1])
            self.__dict__['_crb_prefetch_hits'] = self.__dict__.get('_crb_prefetch_hits', 0) + 1
        if not isinstance(nxt, dict) or 'voxels' in nxt or 'voxel_coords' in nxt:
            return None

        def go():
            from .. import load_data_to_gpu
            nb = dict(nxt)
            nb.pop('_crb_next', None)
            load_data_to_gpu(nb)
            self.prefetch_sparse(nb)
            if nb.get('_vfe_done', False):
                self.__dict__['_crb_prefetched'] = (self._batch_key(nb), nb)
        return go

    def forward(self, batch_dict):
        """training: ({'loss': ..., + training_outputs()}, tb_dict, disp_dict); inference: post_processing's (pred_dicts,
        recall_dicts) — the contract of the reference detectors' forward()"""
        batch_dict = self.run_modules(batch_dict, before_pfe=self._auto_prefetch(batch_dict))
        if not self.training:
            return self.post_processing(batch_dict)
        loss, tb_dict, disp_dict = self.get_training_loss()
        ret = {'loss': loss}
        ret.update(self.training_outputs(batch_dict))
        return ret, tb_dict, disp_dict

    def training_outputs(self, batch_dict):
        """extra entries of the training-mode return dict (detector specific)"""
        return {}

    def build_networks(self):
        ds = self.dataset
        info = {
            'module_list': [],
            'num_rawpoint_features': ds.point_feature_encoder.num_point_features,
            'num_point_features': ds.point_feature_encoder.num_point_features,
            'grid_size': ds.grid_size,
            'point_cloud_range': ds.point_cloud_range,
            'voxel_size': ds.voxel_size,
            'depth_downsample_factor': getattr(ds, 'depth_downsample_factor', None),
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        if getattr(self, 'roi_head', None) is not None and hasattr(getattr(self, 'dense_head', None), 'lazy_box_decode'):
            # two-stage detector: the RPN boxes are read by the RoI head's proposal layer only (its top-k), post-processing
            # reads the RoI head's own boxes — the dense head need not decode all anchors (LAZY_BOX_DECODE = False: it does)
            self.dense_head.lazy_box_decode = LAZY_BOX_DECODE
        from ..backbones_2d.map_to_bev import height_compression
        if height_compression.CHANNELS_LAST:
            import torch as _torch
            for name in ('backbone_2d', 'dense_head'):      # the dense 2-D part runs NHWC end to end
                m = getattr(self, name, None)
                if m is not None:
                    m.to(memory_format=_torch.channels_last)
        return info['module_list']

    def build_vfe(self, model_info_dict):
        cfg = self.model_cfg.get('VFE', None)
        if cfg is None:
            return None, model_info_dict
        m = vfe.__all__[cfg.NAME](
            model_cfg=cfg, num_point_features=model_info_dict['num_rawpoint_features'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            grid_size=model_info_dict['grid_size'], depth_downsample_factor=model_info_dict['depth_downsample_factor'],
            max_num_voxels=getattr(self.dataset, 'max_num_voxels', None),
            max_points_per_voxel=getattr(self.dataset, 'max_points_per_voxel', None))
        model_info_dict['num_point_features'] = m.get_output_feature_dim()
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        cfg = self.model_cfg.get('BACKBONE_3D', None)
        if cfg is None:
            return None, model_info_dict
        m = backbones_3d.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'], grid_size=model_info_dict['grid_size'],
            voxel_size=model_info_dict['voxel_size'], point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['backbone_channels'] = getattr(m, 'backbone_channels', None)
        return m, model_info_dict

    def build_map_to_bev_module(self, model_info_dict):
        cfg = self.model_cfg.get('MAP_TO_BEV', None)
        if cfg is None:
            return None, model_info_dict
        m = map_to_bev.__all__[cfg.NAME](model_cfg=cfg, grid_size=model_info_dict['grid_size'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        cfg = self.model_cfg.get('BACKBONE_2D', None)
        if cfg is None:
            return None, model_info_dict
        m = backbones_2d.__all__[cfg.NAME](model_cfg=cfg, input_channels=model_info_dict['num_bev_features'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_pfe(self, model_info_dict):
        cfg = self.model_cfg.get('PFE', None)
        if cfg is None:
            return None, model_info_dict
        m = pfe.__all__[cfg.NAME](
            model_cfg=cfg, voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'], num_bev_features=model_info_dict['num_bev_features'],
            num_rawpoint_features=model_info_dict['num_rawpoint_features'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['num_point_features_before_fusion'] = m.num_point_features_before_fusion
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        cfg = self.model_cfg.get('DENSE_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        m = dense_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1, class_names=self.class_names,
            grid_size=model_info_dict['grid_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=bool(self.model_cfg.get('ROI_HEAD', False)),
            voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_point_head(self, model_info_dict):
        cfg = self.model_cfg.get('POINT_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        if cfg.get('USE_POINT_FEATURES_BEFORE_FUSION', False):
            c = model_info_dict['num_point_features_before_fusion']
        else:
            c = model_info_dict['num_point_features']
        m = dense_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=c, num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1,
            predict_boxes_when_training=bool(self.model_cfg.get('ROI_HEAD', False)))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_roi_head(self, model_info_dict):
        cfg = self.model_cfg.get('ROI_HEAD', None)
        if cfg is None:
            return None, model_info_dict
        m = roi_heads.__all__[cfg.NAME](
            model_cfg=cfg, input_channels=model_info_dict['num_point_features'],
            backbone_channels=model_info_dict['backbone_channels'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            num_class=self.num_class if not cfg.CLASS_AGNOSTIC else 1)
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def post_processing(self, batch_dict):
        from .post_processing import crb_post_processing
        return crb_post_processing(self, batch_dict)

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        from .post_processing import generate_recall_record
        return generate_recall_record(box_preds, recall_dict, batch_index, data_dict, thresh_list)

    # ------------------------------------------------------------------------------------------------
    def _load_state_dict(self, model_state_disk, *, strict=True):
        """accepts spconv 1.x (k,k,k,Cin,Cout) and 2.x (Cout,k,k,k,Cin) weight layouts
        (detector3d_template.py:455-484)"""
        state_dict = self.state_dict()
        spconv_keys = find_all_spconv_keys(self)
        update = {}
        for key, val in model_state_disk.items():
            if key in spconv_keys and key in state_dict and state_dict[key].shape != val.shape:
                native = val.transpose(-1, -2)
                if native.shape == state_dict[key].shape:
                    val = native.contiguous()
                else:
                    assert val.dim() == 5, 'currently only spconv 3D is supported'
                    implicit = val.permute(4, 0, 1, 2, 3)
                    if implicit.shape == state_dict[key].shape:
                        val = implicit.contiguous()
            if key in state_dict and state_dict[key].shape == val.shape:
                update[key] = val
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update

    def load_params_from_file(self, filename, logger, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if to_cpu else 'GPU'))
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None)
        state_dict, update = self._load_state_dict(checkpoint['model_state'], strict=False)
        for key in state_dict:
            if key not in update:
                logger.info('Not updated weight %s: %s' % (key, str(state_dict[key].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update), len(state_dict)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError
        checkpoint = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None)
        epoch, it = checkpoint.get('epoch', -1), checkpoint.get('it', 0.0)
        self._load_state_dict(checkpoint['model_state'], strict=True)
        if optimizer is not None and checkpoint.get('optimizer_state', None) is not None:
            optimizer.load_state_dict(checkpoint['optimizer_state'])
        return it, epoch
