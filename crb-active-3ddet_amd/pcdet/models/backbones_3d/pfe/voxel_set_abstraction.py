"""VoxelSetAbstraction (pcdet/models/backbones_3d/pfe/voxel_set_abstraction.py:124-411): keypoint sampling (FPS, HIP),
bilinear BEV feature lookup, multi-scale set abstraction over raw points and the sparse-conv feature volumes.

Host-side differences from the reference: the per-frame Python loops (FPS per frame :250-256, bs_mask loops :346-347,
:176-204) are replaced by batched calls — all frames share ONE FPS launch when they have the same point count, and the
per-frame counts come from one bincount instead of `(idx == k).sum()` round trips."""
import torch
import torch.nn as nn

from ....ops.pointnet2.pointnet2_stack import pointnet2_modules as pointnet2_stack_modules
from ....ops.pointnet2.pointnet2_stack import pointnet2_utils as pointnet2_stack_utils
from ....utils import common_utils
from ....utils.fc_rows import fc_rows


def bilinear_interpolate_torch(im, x, y):
    """im (H,W,C), x (N), y (N) -> (N,C)  (voxel_set_abstraction.py:11-42)"""
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0 = torch.clamp(x0, 0, im.shape[1] - 1)
    x1 = torch.clamp(x1, 0, im.shape[1] - 1)
    y0 = torch.clamp(y0, 0, im.shape[0] - 1)
    y1 = torch.clamp(y1, 0, im.shape[0] - 1)
    wa = (x1.type_as(x) - x) * (y1.type_as(y) - y)
    wb = (x1.type_as(x) - x) * (y - y0.type_as(y))
    wc = (x - x0.type_as(x)) * (y1.type_as(y) - y)
    wd = (x - x0.type_as(x)) * (y - y0.type_as(y))
    return (im[y0, x0] * wa[:, None] + im[y1, x0] * wb[:, None] + im[y0, x1] * wc[:, None] + im[y1, x1] * wd[:, None])


BEV_INTERP_KERNEL = True   # crb_bev_interpolate_forward / _backward (one launch each) instead of the torch expression below


CU_RESERVATION = __import__('os').environ.get('CRB_CU_RESERVATION', '1') == '1'    # A/B: announce the CUs the sampling holds (crb_cu_reservation)

class _BevInterpolate(torch.autograd.Function):
    """bilinear lookup of the channels_last BEV map at the keypoints: same operations in the same order as the torch expression of
    interpolate_from_bev_features (bit-identical forward); backward adds the four weighted copies of the gradient into the map
    gradient with float atomics (the torch path: four sort-based index_put calls, 35 launches per step)"""

    @staticmethod
    def forward(ctx, bev, keypoints, x_min, y_min, vx, vy, stride):
        from crbhip import lib, check, ptr, cur_stream, require_cuda
        require_cuda(bev, keypoints)
        if not bev.is_contiguous(memory_format=torch.channels_last):
            bev = bev.contiguous(memory_format=torch.channels_last)
        B, C, H, W = bev.shape
        M = keypoints.shape[0]
        out = torch.empty((M, C), dtype=torch.float32, device=bev.device)
        check(lib.crb_bev_interpolate_forward(bev.data_ptr(), B, H, W, C, ptr(keypoints), M, x_min, y_min, vx, vy, stride, ptr(out),
                                              cur_stream(bev.device)), 'crb_bev_interpolate_forward')
        ctx.save_for_backward(keypoints)
        ctx.geom = (B, C, H, W, x_min, y_min, vx, vy, stride)
        return out

    @staticmethod
    def backward(ctx, dout):
        from crbhip import lib, check, ptr, cur_stream
        keypoints, = ctx.saved_tensors
        B, C, H, W, x_min, y_min, vx, vy, stride = ctx.geom
        dout = dout.contiguous().float()
        dbev = torch.zeros((B, H, W, C), dtype=torch.float32, device=dout.device).permute(0, 3, 1, 2)       # channels_last (B,C,H,W)
        check(lib.crb_bev_interpolate_backward(ptr(dout), B, H, W, C, ptr(keypoints), keypoints.shape[0], x_min, y_min, vx, vy, stride,
                                               dbev.data_ptr(), cur_stream(dout.device)), 'crb_bev_interpolate_backward')
        return dbev, None, None, None, None, None, None


def _batch_counts(bs_idx, batch_size):
    return common_utils.batch_counts(bs_idx, batch_size)


class VoxelSetAbstraction(nn.Module):
    def __init__(self, model_cfg, voxel_size, point_cloud_range, num_bev_features=None, num_rawpoint_features=None,
                 **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        SA_cfg = self.model_cfg.SA_LAYER
        self.SA_layers = nn.ModuleList()
        self.SA_layer_names = []
        self.downsample_times_map = {}
        c_in = 0
        for src_name in self.model_cfg.FEATURES_SOURCE:
            if src_name in ['bev', 'raw_points']:
                continue
            self.downsample_times_map[src_name] = SA_cfg[src_name].DOWNSAMPLE_FACTOR
            if SA_cfg[src_name].get('INPUT_CHANNELS', None) is None:
                first = SA_cfg[src_name].MLPS[0]
                input_channels = first[0] if isinstance(first, (list, tuple)) else first
            else:
                input_channels = SA_cfg[src_name]['INPUT_CHANNELS']
            layer, c_out = pointnet2_stack_modules.build_local_aggregation_module(input_channels=input_channels,
                                                                                  config=SA_cfg[src_name])
            self.SA_layers.append(layer)
            self.SA_layer_names.append(src_name)
            c_in += c_out
        if 'bev' in self.model_cfg.FEATURES_SOURCE:
            c_in += num_bev_features
        if 'raw_points' in self.model_cfg.FEATURES_SOURCE:
            self.SA_rawpoints, c_out = pointnet2_stack_modules.build_local_aggregation_module(
                input_channels=num_rawpoint_features - 3, config=SA_cfg['raw_points'])
            c_in += c_out
        self.vsa_point_feature_fusion = nn.Sequential(
            nn.Linear(c_in, self.model_cfg.NUM_OUTPUT_FEATURES, bias=False),
            nn.BatchNorm1d(self.model_cfg.NUM_OUTPUT_FEATURES), nn.ReLU())
        self.num_point_features = self.model_cfg.NUM_OUTPUT_FEATURES
        self.num_point_features_before_fusion = c_in

    def interpolate_from_bev_features(self, keypoints, bev_features, batch_size, bev_stride):
        """keypoints (M,4) [b,x,y,z], bev (B,C,H,W) -> (M,C); one gather over all frames"""
        # (the kernel's backward adds with float atomics in arrival order: under torch.use_deterministic_algorithms the torch
        # expression below - sort-based index_put in backward, bit-reproducible - is what runs)
        det = torch.are_deterministic_algorithms_enabled() and torch.is_grad_enabled() and bev_features.requires_grad
        if BEV_INTERP_KERNEL and not det and bev_features.is_cuda and bev_features.dtype == torch.float32 and \
                bev_features.shape[1] % 4 == 0 and keypoints.dtype == torch.float32:
            return _BevInterpolate.apply(bev_features, keypoints.contiguous(), float(self.point_cloud_range[0]),
                                         float(self.point_cloud_range[1]), float(self.voxel_size[0]), float(self.voxel_size[1]),
                                         float(bev_stride))
        x = (keypoints[:, 1] - self.point_cloud_range[0]) / self.voxel_size[0] / bev_stride
        y = (keypoints[:, 2] - self.point_cloud_range[1]) / self.voxel_size[1] / bev_stride
        B, C, H, W = bev_features.shape
        b = keypoints[:, 0].long()
        im = bev_features.permute(0, 2, 3, 1).reshape(B * H, W, C)       # frames stacked along y
        x0 = torch.floor(x).long()
        y0 = torch.floor(y).long()
        x1, y1 = x0 + 1, y0 + 1
        x0c, x1c = torch.clamp(x0, 0, W - 1), torch.clamp(x1, 0, W - 1)
        y0c, y1c = torch.clamp(y0, 0, H - 1), torch.clamp(y1, 0, H - 1)
        wa = (x1c.type_as(x) - x) * (y1c.type_as(y) - y)
        wb = (x1c.type_as(x) - x) * (y - y0c.type_as(y))
        wc = (x - x0c.type_as(x)) * (y1c.type_as(y) - y)
        wd = (x - x0c.type_as(x)) * (y - y0c.type_as(y))
        r0, r1 = b * H + y0c, b * H + y1c
        return (im[r0, x0c] * wa[:, None] + im[r1, x0c] * wb[:, None] + im[r0, x1c] * wc[:, None] +
                im[r1, x1c] * wd[:, None])

    def get_sampled_points(self, batch_dict):
        """-> keypoints (B*K, 4) [bs_idx, x, y, z]  (voxel_set_abstraction.py:224-275, SAMPLE_METHOD FPS)"""
        batch_size = batch_dict['batch_size']
        K = self.model_cfg.NUM_KEYPOINTS
        if self.model_cfg.POINT_SOURCE == 'raw_points':
            src_points = batch_dict['points'][:, 1:4]
            batch_indices = batch_dict['points'][:, 0].long()
        elif self.model_cfg.POINT_SOURCE == 'voxel_centers':
            src_points = common_utils.get_voxel_centers(batch_dict['voxel_coords'][:, 1:4], downsample_times=1,
                                                        voxel_size=self.voxel_size,
                                                        point_cloud_range=self.point_cloud_range)
            batch_indices = batch_dict['voxel_coords'][:, 0].long()
        else:
            raise NotImplementedError
        if self.model_cfg.SAMPLE_METHOD != 'FPS':
            raise NotImplementedError('SPC sampling is PV-RCNN++ only (out of scope)')
        counts = batch_dict.get('point_frame_counts_host', None)
        if counts is None:
            counts = torch.bincount(batch_indices, minlength=batch_size).tolist()
        if len(set(counts)) == 1 and counts[0] >= K:
            # equal-length frames: one launch, one workgroup per frame
            xyz = src_points.reshape(batch_size, counts[0], 3).contiguous()
            idx = pointnet2_stack_utils.farthest_point_sample(xyz, K).long()
            keypoints = torch.gather(xyz, 1, idx[..., None].expand(-1, -1, 3))
        else:
            kps, s = [], 0
            for n in counts:
                pts = src_points[s:s + n].unsqueeze(0).contiguous()
                cur = pointnet2_stack_utils.farthest_point_sample(pts, K).long()
                if n < K:
                    times = int(K / n) + 1
                    cur[0] = cur[0, :n].repeat(times)[:K]
                kps.append(pts[0][cur[0]].unsqueeze(0))
                s += n
            keypoints = torch.cat(kps, dim=0)
        batch_idx = torch.arange(batch_size, device=keypoints.device).view(-1, 1).repeat(1, K).view(-1, 1)
        return torch.cat((batch_idx.float(), keypoints.reshape(-1, 3)), dim=1)

    @staticmethod
    def aggregate_keypoint_features_from_one_source(batch_size, aggregate_func, xyz, xyz_features, xyz_bs_idxs, new_xyz,
                                                    new_xyz_batch_cnt, **unused):
        xyz_batch_cnt = _batch_counts(xyz_bs_idxs, batch_size)
        _, pooled = aggregate_func(xyz=xyz.contiguous(), xyz_batch_cnt=xyz_batch_cnt, new_xyz=new_xyz,
                                   new_xyz_batch_cnt=new_xyz_batch_cnt, features=xyz_features.contiguous())
        return pooled

    PREFETCH_KEYPOINTS = True

    def prefetch_keypoints(self, batch_dict):
        """Start farthest-point sampling on a side HIP stream as soon as the raw points exist. FPS is a sequential
        2048-round selection that keeps one workgroup per frame busy for ~5 ms and depends on nothing but the points, so
        it runs under the 3-D / BEV backbones (which use the other 240 CUs) instead of after them. The detector calls
        this before its module loop; forward() joins the stream."""
        pts = batch_dict.get('points', None)
        if not self.PREFETCH_KEYPOINTS or pts is None or not pts.is_cuda or self.model_cfg.POINT_SOURCE != 'raw_points':
            return
        main = torch.cuda.current_stream(pts.device)
        side = getattr(self, '_kp_stream', None)
        if side is None or side.device != pts.device:
            side = self._kp_stream = torch.cuda.Stream(device=pts.device)
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            from crbhip import lib, check, cur_stream
            # one CU per frame is taken for the ~5 ms of the sampling: the persistent Winograd launches of the BEV backbone on the main
            # stream spread their units over the other CUs meanwhile (crb_cu_reservation; scoring at 16 frames per batch: 0.92 ->
            # 0.78 ms per convolution while the sampling runs)
            # (only the farthest-point sampler holds CUs for milliseconds; at most a quarter of the device is announced - the kernel
            # itself lets at most half of a launch's workgroups give way)
            fps = self.model_cfg.get('SAMPLE_METHOD', 'FPS') == 'FPS' and CU_RESERVATION
            held = min(int(batch_dict['batch_size']), torch.cuda.get_device_properties(pts.device).multi_processor_count // 4)
            if fps:
                check(lib.crb_cu_reservation(held, cur_stream(pts.device)), 'crb_cu_reservation')
            try:
                kp = self.get_sampled_points(batch_dict)
            finally:                                  # (a reservation left standing costs speed only, never results)
                if fps:
                    check(lib.crb_cu_reservation(0, cur_stream(pts.device)), 'crb_cu_reservation')
            done = torch.cuda.Event()
            done.record(side)
        pts.record_stream(side)
        batch_dict['_keypoints_prefetched'] = (kp, done)

    def forward(self, batch_dict):
        pre = batch_dict.pop('_keypoints_prefetched', None)
        if pre is not None:
            keypoints, done = pre
            cur = torch.cuda.current_stream(keypoints.device)
            cur.wait_event(done)
            keypoints.record_stream(cur)
        else:
            keypoints = self.get_sampled_points(batch_dict)
        batch_size = batch_dict['batch_size']
        feats = []
        if 'bev' in self.model_cfg.FEATURES_SOURCE:
            feats.append(self.interpolate_from_bev_features(keypoints, batch_dict['spatial_features'], batch_size,
                                                            bev_stride=batch_dict['spatial_features_stride']))
        new_xyz = keypoints[:, 1:4].contiguous()
        # (get_sampled_points returns exactly NUM_KEYPOINTS rows per frame, frames in order; any other keypoint source is counted)
        if keypoints.shape[0] == batch_size * self.model_cfg.NUM_KEYPOINTS:
            new_xyz_batch_cnt = torch.full((batch_size,), self.model_cfg.NUM_KEYPOINTS, dtype=torch.int32, device=keypoints.device)
        else:
            new_xyz_batch_cnt = _batch_counts(keypoints[:, 0], batch_size)
        if 'raw_points' in self.model_cfg.FEATURES_SOURCE:
            raw = batch_dict['points']
            feats.append(self.aggregate_keypoint_features_from_one_source(
                batch_size=batch_size, aggregate_func=self.SA_rawpoints, xyz=raw[:, 1:4],
                xyz_features=raw[:, 4:].contiguous(), xyz_bs_idxs=raw[:, 0], new_xyz=new_xyz,
                new_xyz_batch_cnt=new_xyz_batch_cnt))
        for k, src_name in enumerate(self.SA_layer_names):
            sp = batch_dict['multi_scale_3d_features'][src_name]
            xyz = common_utils.get_voxel_centers(sp.indices[:, 1:4], downsample_times=self.downsample_times_map[src_name],
                                                 voxel_size=self.voxel_size, point_cloud_range=self.point_cloud_range)
            feats.append(self.aggregate_keypoint_features_from_one_source(
                batch_size=batch_size, aggregate_func=self.SA_layers[k], xyz=xyz.contiguous(),
                xyz_features=sp.features.contiguous(), xyz_bs_idxs=sp.indices[:, 0], new_xyz=new_xyz,
                new_xyz_batch_cnt=new_xyz_batch_cnt))
        point_features = torch.cat(feats, dim=-1)
        batch_dict['point_features_before_fusion'] = point_features.view(-1, point_features.shape[-1])
        batch_dict['point_features'] = fc_rows(self.vsa_point_feature_fusion, point_features.view(-1, point_features.shape[-1]))
        batch_dict['point_coords'] = keypoints
        return batch_dict
