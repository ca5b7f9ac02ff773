"""Synthetic KITTI / Waymo shaped dataset exposing what Detector3DTemplate.build_networks and Strategy read
(SURVEY Appendix D): class_names, point_feature_encoder.num_point_features, grid_size, point_cloud_range, voxel_size,
depth_downsample_factor, sample_id_list / kitti_infos, collate_batch.

Two batch layouts:
  device_voxelize=True  (default, MI355X path): batches carry raw points only; MeanVFE runs the HIP voxel generator.
  device_voxelize=False (reference layout): each frame is voxelized through VoxelGeneratorWrapper (HIP kernel behind
                         the spconv.utils.Point2VoxelCPU3d surface) and collated exactly like the reference."""
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import synthetic as syn
from .sampler import DistributedSampler


class _PointFeatureEncoder(object):
    def __init__(self, n):
        self.num_point_features = n


class SyntheticDataset(Dataset):
    def __init__(self, num_frames=64, n_points=20000, kind='kitti', training=True, first_frame=0,
                 device_voxelize=True, class_names=None):
        assert kind in ('kitti', 'waymo')
        self.kind, self.training = kind, training
        self.n_points, self.first_frame = n_points, first_frame
        self.device_voxelize = device_voxelize
        if kind == 'kitti':
            self.point_cloud_range = np.array(syn.KITTI_RANGE, dtype=np.float32)
            self.voxel_size = list(syn.KITTI_VOXEL)
            self.class_names = class_names or ['Car', 'Pedestrian', 'Cyclist']
            self.max_num_voxels = {'train': 16000, 'test': 40000}
            nfeat = 4
        else:
            self.point_cloud_range = np.array(syn.WAYMO_RANGE, dtype=np.float32)
            self.voxel_size = list(syn.WAYMO_VOXEL)
            self.class_names = class_names or ['Vehicle', 'Pedestrian', 'Cyclist']
            self.max_num_voxels = {'train': 150000, 'test': 150000}
            nfeat = 5
        self.max_points_per_voxel = 5
        self.point_feature_encoder = _PointFeatureEncoder(nfeat)
        g = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(self.voxel_size)
        self.grid_size = np.round(g).astype(np.int64)
        self.depth_downsample_factor = None
        self.sample_id_list = ['%06d' % (first_frame + i) for i in range(num_frames)]
        self.kitti_infos = [{'point_cloud': {'lidar_idx': s}, 'frame_id': s} for s in self.sample_id_list]
        self.frame_ids = self.sample_id_list
        self.infos = self.kitti_infos
        self._voxel_generator = None

    def sync_id_views(self, waymo=False):
        """the active loop re-assigns (sample_id_list, kitti_infos) for KITTI or (frame_ids, infos) for Waymo
        (pcdet/datasets/__init__.py:111-147); keep the other pair of names pointing at the same lists"""
        if waymo:
            self.sample_id_list, self.kitti_infos = self.frame_ids, self.infos
        else:
            self.frame_ids, self.infos = self.sample_id_list, self.kitti_infos

    @property
    def mode(self):
        return 'train' if self.training else 'test'

    def __len__(self):
        return len(self.sample_id_list)

    def __getitem__(self, index):
        fid = self.sample_id_list[index]
        pts, boxes = syn.kitti_frame(int(fid), self.n_points, waymo=(self.kind == 'waymo'))
        d = {'points': pts, 'gt_boxes': boxes, 'frame_id': fid, 'use_lead_xyz': True}
        if not self.device_voxelize:
            from .processor.data_processor import VoxelGeneratorWrapper
            if self._voxel_generator is None:
                self._voxel_generator = VoxelGeneratorWrapper(
                    vsize_xyz=self.voxel_size, coors_range_xyz=self.point_cloud_range,
                    num_point_features=self.point_feature_encoder.num_point_features,
                    max_num_points_per_voxel=self.max_points_per_voxel,
                    max_num_voxels=self.max_num_voxels[self.mode])
            v, c, n = self._voxel_generator.generate(pts)
            d.update({'voxels': v, 'voxel_coords': c, 'voxel_num_points': n})
        return d

    @staticmethod
    def collate_batch(batch_list, _unused=False):
        """same key layout as DatasetTemplate.collate_batch (pcdet/datasets/dataset.py:160-229)"""
        data = defaultdict(list)
        for s in batch_list:
            for k, v in s.items():
                data[k].append(v)
        B = len(batch_list)
        ret = {}
        for key, val in data.items():
            if key in ('voxels', 'voxel_num_points'):
                ret[key] = np.concatenate(val, axis=0)
            elif key in ('points', 'voxel_coords'):
                ret[key] = np.concatenate([np.pad(c, ((0, 0), (1, 0)), mode='constant', constant_values=i)
                                           for i, c in enumerate(val)], axis=0)
                if key == 'points':
                    ret['point_frame_offsets'] = np.concatenate([[0], np.cumsum([len(c) for c in val])]).astype(np.int32)
            elif key == 'gt_boxes':
                mx = max(len(x) for x in val)
                g = np.zeros((B, mx, val[0].shape[-1]), dtype=np.float32)
                for k in range(B):
                    g[k, :len(val[k])] = val[k]
                ret[key] = g
            elif key == 'frame_id':
                ret[key] = np.array(val)
            else:
                ret[key] = np.stack(val, axis=0)
        ret['batch_size'] = B
        return ret


def to_device_batch(batch, device):
    """host batch -> device tensors with the dtypes the HIP path wants (offsets int32, the rest float32)"""
    out = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray) and k == 'point_frame_offsets':
            out[k] = torch.from_numpy(v).to(device=device, dtype=torch.int32)
        elif isinstance(v, np.ndarray) and v.dtype.kind in 'fiu' and k not in ('frame_id',):
            out[k] = torch.from_numpy(v).float().to(device)
        else:
            out[k] = v
    return out


def build_synthetic_dataloader(dataset, batch_size, dist=False, workers=0, shuffle=False, rank=None, world=None):
    sampler = DistributedSampler(dataset, world, rank, shuffle=shuffle) if dist else None
    return DataLoader(dataset, batch_size=batch_size, pin_memory=True, num_workers=workers,
                      shuffle=(sampler is None) and shuffle, collate_fn=dataset.collate_batch, drop_last=False,
                      sampler=sampler, timeout=0)
