"""VoxelGeneratorWrapper (pcdet/datasets/processor/data_processor.py:15-60) over the gfx950 voxel generator."""
import numpy as np


class VoxelGeneratorWrapper():
    def __init__(self, vsize_xyz, coors_range_xyz, num_point_features, max_num_points_per_voxel, max_num_voxels):
        from spconv.utils import Point2VoxelCPU3d as VoxelGenerator
        self.spconv_ver = 2
        self._voxel_generator = VoxelGenerator(
            vsize_xyz=vsize_xyz, coors_range_xyz=coors_range_xyz, num_point_features=num_point_features,
            max_num_points_per_voxel=max_num_points_per_voxel, max_num_voxels=max_num_voxels)

    def generate(self, points):
        """points (n,C) float32 numpy -> voxels (M,T,C), coordinates (M,3)[z,y,x], num_points (M)"""
        import cumm.tensorview as tv
        tv_voxels, tv_coordinates, tv_num_points = self._voxel_generator.point_to_voxel(
            tv.from_numpy(np.ascontiguousarray(points, dtype=np.float32)))
        return tv_voxels.numpy(), tv_coordinates.numpy(), tv_num_points.numpy()


class DataProcessor(object):
    """Config-driven per-frame processor queue (pcdet/datasets/processor/data_processor.py:62-143), host arrays in and out:
    mask_points_and_boxes_outside_range, shuffle_points, transform_points_to_voxels(_placeholder). Voxelisation goes
    through VoxelGeneratorWrapper (the gfx950 voxel generator behind spconv's Point2VoxelCPU3d interface).
    `sample_points` belongs to the point-based detectors (PointRCNN) and is not provided."""

    def __init__(self, processor_configs, point_cloud_range, training, num_point_features):
        from functools import partial
        self._partial = partial
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.num_point_features = num_point_features
        self.mode = 'train' if training else 'test'
        self.grid_size = self.voxel_size = None
        self.voxel_generator = None
        self.data_processor_queue = []
        for cur_cfg in processor_configs:
            if not hasattr(self, cur_cfg.NAME):
                raise NotImplementedError('DATA_PROCESSOR step %s' % cur_cfg.NAME)
            self.data_processor_queue.append(getattr(self, cur_cfg.NAME)(config=cur_cfg))

    def mask_points_and_boxes_outside_range(self, data_dict=None, config=None):
        if data_dict is None:
            return self._partial(self.mask_points_and_boxes_outside_range, config=config)
        from ...utils import box_utils, common_utils
        if data_dict.get('points', None) is not None:
            mask = common_utils.mask_points_by_range(data_dict['points'], self.point_cloud_range)
            data_dict['points'] = data_dict['points'][mask]
        if data_dict.get('gt_boxes', None) is not None and config.REMOVE_OUTSIDE_BOXES and self.training:
            mask = box_utils.mask_boxes_outside_range_numpy(data_dict['gt_boxes'], self.point_cloud_range,
                                                            min_num_corners=config.get('min_num_corners', 1))
            data_dict['gt_boxes'] = data_dict['gt_boxes'][mask]
        return data_dict

    def shuffle_points(self, data_dict=None, config=None):
        if data_dict is None:
            return self._partial(self.shuffle_points, config=config)
        if config.SHUFFLE_ENABLED[self.mode]:
            points = data_dict['points']
            data_dict['points'] = points[np.random.permutation(points.shape[0])]
        return data_dict

    def _bind_grid(self, config):
        grid_size = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(config.VOXEL_SIZE)
        self.grid_size = np.round(grid_size).astype(np.int64)
        self.voxel_size = config.VOXEL_SIZE

    def transform_points_to_voxels_placeholder(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_grid(config)
            return self._partial(self.transform_points_to_voxels_placeholder, config=config)
        return data_dict

    def transform_points_to_voxels(self, data_dict=None, config=None):
        if data_dict is None:
            self._bind_grid(config)
            return self._partial(self.transform_points_to_voxels, config=config)
        if self.voxel_generator is None:
            self.voxel_generator = VoxelGeneratorWrapper(
                vsize_xyz=config.VOXEL_SIZE, coors_range_xyz=self.point_cloud_range,
                num_point_features=self.num_point_features, max_num_points_per_voxel=config.MAX_POINTS_PER_VOXEL,
                max_num_voxels=config.MAX_NUMBER_OF_VOXELS[self.mode])
        voxels, coordinates, num_points = self.voxel_generator.generate(data_dict['points'])
        if not data_dict['use_lead_xyz']:
            voxels = voxels[..., 3:]
        data_dict['voxels'] = voxels
        data_dict['voxel_coords'] = coordinates
        data_dict['voxel_num_points'] = num_points
        return data_dict

    def forward(self, data_dict):
        for cur_processor in self.data_processor_queue:
            data_dict = cur_processor(data_dict=data_dict)
        return data_dict


class DeviceDataProcessor(object):
    """the same queue for a whole BATCH on the GPU. Raw per-frame point arrays cross PCIe once (pinned,
    non-blocking); range mask, per-frame shuffle and frame concatenation run as device ops, voxelisation is left to
    MeanVFE's crb_voxelize call (points + point_frame_offsets in the batch, no (M,5,C) voxel tensor at all). GT boxes are a
    few dozen rows per frame and stay on the host path.

    Differences from the host queue: the shuffle draws from a torch device generator, not np.random (another permutation
    of the same points: 'first 5 points per voxel / first M voxels' pick different but equally valid members)."""

    def __init__(self, processor_configs, point_cloud_range, training, num_point_features, device='cuda'):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.point_cloud_range = np.asarray(point_cloud_range, dtype=np.float32)
        self.training = training
        self.mode = 'train' if training else 'test'
        self.num_point_features = num_point_features
        self.mask_cfg = self.shuffle = None
        self.voxel_cfg = None
        for cfg in processor_configs:
            if cfg.NAME == 'mask_points_and_boxes_outside_range':
                self.mask_cfg = cfg
            elif cfg.NAME == 'shuffle_points':
                self.shuffle = bool(cfg.SHUFFLE_ENABLED[self.mode])
            elif cfg.NAME in ('transform_points_to_voxels', 'transform_points_to_voxels_placeholder'):
                self.voxel_cfg = cfg
            else:
                raise NotImplementedError('DATA_PROCESSOR step %s' % cfg.NAME)
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(0)
        if self.voxel_cfg is not None:
            g = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(self.voxel_cfg.VOXEL_SIZE)
            self.grid_size = np.round(g).astype(np.int64)
            self.voxel_size = self.voxel_cfg.VOXEL_SIZE

    def process_batch(self, points_list, gt_boxes_list=None, frame_ids=None):
        """points_list: per-frame (n_i, C) float32 numpy arrays -> batch dict with device 'points' (N,1+C),
        'point_frame_offsets' (B+1) int32, host-padded 'gt_boxes' (B,G,8) on the device"""
        torch = self.torch
        from ...utils import box_utils, common_utils
        B = len(points_list)
        counts = [len(p) for p in points_list]
        host = torch.from_numpy(np.concatenate(points_list, 0).astype(np.float32, copy=False))
        pts = (host.pin_memory() if self.device.type == 'cuda' else host).to(self.device, non_blocking=True)
        bidx = torch.repeat_interleave(torch.arange(B, device=self.device),
                                       torch.tensor(counts, device=self.device))
        if self.mask_cfg is not None:
            r = self.point_cloud_range
            keep = (pts[:, 0] >= float(r[0])) & (pts[:, 0] <= float(r[3])) & (pts[:, 1] >= float(r[1])) & \
                   (pts[:, 1] <= float(r[4]))
            pts, bidx = pts[keep], bidx[keep]                   # the one device->host size read-back of the batch
        if self.shuffle:
            key = bidx.double() + torch.rand(bidx.shape[0], device=self.device, generator=self.generator,
                                             dtype=torch.float64)
            order = torch.argsort(key)
            pts = pts[order]
        n_per = common_utils.batch_counts(bidx, B)
        off = torch.zeros((B + 1,), dtype=torch.int32, device=self.device)
        off[1:] = torch.cumsum(n_per, 0)
        batch = {'points': torch.cat([bidx.float().unsqueeze(1), pts], 1), 'point_frame_offsets': off, 'batch_size': B}
        if gt_boxes_list is not None:
            gts = []
            for g in gt_boxes_list:
                g = np.asarray(g, dtype=np.float32)
                if self.mask_cfg is not None and self.mask_cfg.REMOVE_OUTSIDE_BOXES and self.training and len(g):
                    g = g[box_utils.mask_boxes_outside_range_numpy(g, self.point_cloud_range,
                                                                   self.mask_cfg.get('min_num_corners', 1))]
                gts.append(g)
            mx = max(1, max(len(g) for g in gts))
            width = gts[0].shape[-1] if len(gts[0].shape) == 2 and gts[0].shape[-1] else 8
            pad = np.zeros((B, mx, width), dtype=np.float32)
            for k, g in enumerate(gts):
                pad[k, :len(g)] = g
            batch['gt_boxes'] = torch.from_numpy(pad).to(self.device, non_blocking=True)
        if frame_ids is not None:
            batch['frame_id'] = np.array(frame_ids)
        return batch
