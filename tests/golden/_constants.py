"""Constants and seeded input builders shared by the golden-vector generator (make_goldens.py, which imports the reference from
/root/reference and stays in this container) and the tests that regenerate the same inputs. No reference import here: this module
travels to the GPU box, the generator does not have to."""
import numpy as np
import torch


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {})
        d.update(kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, list):
            v = [EasyDict(x) if isinstance(x, dict) else x for x in v]
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def small_head_cfg():
    def anc(cls, size, bottom, m, u):
        return {'class_name': cls, 'anchor_sizes': [size], 'anchor_rotations': [0, 1.57],
                'anchor_bottom_heights': [bottom], 'align_center': False, 'feature_map_stride': 8,
                'matched_threshold': m, 'unmatched_threshold': u}
    return EasyDict({
        'NAME': 'AnchorHeadSingle', 'CLASS_AGNOSTIC': False, 'USE_DIRECTION_CLASSIFIER': True,
        'DIR_OFFSET': 0.78539, 'DIR_LIMIT_OFFSET': 0.0, 'NUM_DIR_BINS': 2,
        'ANCHOR_GENERATOR_CONFIG': [anc('Car', [3.9, 1.6, 1.56], -1.78, 0.6, 0.45),
                                    anc('Pedestrian', [0.8, 0.6, 1.73], -0.6, 0.5, 0.35),
                                    anc('Cyclist', [1.76, 0.6, 1.73], -0.6, 0.5, 0.35)],
        'TARGET_ASSIGNER_CONFIG': {'NAME': 'AxisAlignedTargetAssigner', 'POS_FRACTION': -1.0, 'SAMPLE_SIZE': 512,
                                   'NORM_BY_NUM_EXAMPLES': False, 'MATCH_HEIGHT': False, 'BOX_CODER': 'ResidualCoder'},
        'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0, 'loc_weight': 2.0, 'dir_weight': 0.2,
                                         'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}},
    })


def seeded_state(module, seed):
    """parameters and buffers of a module filled from ONE numpy stream in name order (weights are regenerated the same way
    by the test instead of being stored): conv / linear weights ~ N(0, 1/sqrt(fan_in)), BN weight in [0.5, 1.5], BN bias
    and running_mean ~ N(0, 0.2), running_var in [0.5, 1.5]"""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, v in sorted(module.state_dict().items()):
        shape = tuple(v.shape)
        if name.endswith('num_batches_tracked'):
            a = np.zeros(shape, np.int64)
        elif name.endswith('running_var') or (name.endswith('weight') and len(shape) == 1):
            a = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif len(shape) == 1:
            a = rng.normal(0, 0.2, shape).astype(np.float32)
        else:
            fan_in = int(np.prod(shape)) // shape[0]
            a = (rng.normal(0, 1, shape) / np.sqrt(fan_in)).astype(np.float32)
        sd[name] = torch.from_numpy(a)
    return sd


def parta2_cfg():
    return EasyDict({
        'NAME': 'PartA2FCHead', 'CLASS_AGNOSTIC': True, 'SHARED_FC': [48, 48], 'CLS_FC': [24], 'REG_FC': [24], 'DP_RATIO': 0.0,
        'DISABLE_PART': False, 'SEG_MASK_SCORE_THRESH': 0.3,
        'NMS_CONFIG': {'TRAIN': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 64,
                                 'NMS_POST_MAXSIZE': 16, 'NMS_THRESH': 0.8},
                       'TEST': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 64,
                                'NMS_POST_MAXSIZE': 16, 'NMS_THRESH': 0.7}},
        'ROI_AWARE_POOL': {'POOL_SIZE': 4, 'NUM_FEATURES': 64, 'MAX_POINTS_PER_VOXEL': 32},
        'TARGET_CONFIG': {'BOX_CODER': 'ResidualCoder', 'ROI_PER_IMAGE': 16, 'FG_RATIO': 0.5, 'SAMPLE_ROI_BY_EACH_CLASS': True,
                          'CLS_SCORE_TYPE': 'roi_iou', 'CLS_FG_THRESH': 0.75, 'CLS_BG_THRESH': 0.25,
                          'CLS_BG_THRESH_LO': 0.1, 'HARD_BG_RATIO': 0.8, 'REG_FG_THRESH': 0.55},
        'LOSS_CONFIG': {'CLS_LOSS': 'BinaryCrossEntropy', 'REG_LOSS': 'smooth-l1', 'CORNER_LOSS_REGULARIZATION': True,
                        'LOSS_WEIGHTS': {'rcnn_cls_weight': 1.0, 'rcnn_reg_weight': 1.0, 'rcnn_corner_weight': 1.0,
                                         'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}},
    })


def parta2_inputs(seed=41, B=2, R=16, C=32):
    """per frame: 5 objects with 40-120 points each + clutter; 16 RoIs = jittered object boxes, far-away empty boxes and two
    all-zero rows (the padding proposal_layer leaves)"""
    rng = np.random.default_rng(seed)
    sizes = [(3.9, 1.6, 1.56), (0.8, 0.6, 1.73), (1.76, 0.6, 1.73)]
    pts, rois = [], np.zeros((B, R, 7), np.float32)
    for b in range(B):
        obj = []
        for g in range(5):
            s = np.array(sizes[g % 3]) * rng.uniform(0.9, 1.1, 3)
            obj.append([rng.uniform(5, 60), rng.uniform(-30, 30), rng.uniform(-1.2, -0.6), *s, rng.uniform(-np.pi, np.pi)])
        obj = np.array(obj, np.float32)
        p = [np.stack([rng.uniform(0, 70, 300), rng.uniform(-40, 40, 300), rng.uniform(-3, 1, 300)], 1)]
        for o in obj:
            k = int(rng.integers(40, 120))
            loc = rng.uniform(-0.55, 0.55, (k, 3)) * o[3:6]
            ca, sa = np.cos(o[6]), np.sin(o[6])
            p.append(np.stack([loc[:, 0] * ca - loc[:, 1] * sa + o[0], loc[:, 0] * sa + loc[:, 1] * ca + o[1], loc[:, 2] + o[2]], 1))
        p = np.concatenate(p).astype(np.float32)
        pts.append(np.concatenate([np.full((len(p), 1), b, np.float32), p], 1))
        for r in range(R - 2):
            if r < 11:
                o = obj[r % 5].copy()
                o[:3] += rng.normal(0, 0.25, 3)
                o[3:6] *= rng.uniform(0.9, 1.2, 3)
                o[6] += rng.normal(0, 0.15)
                rois[b, r] = o
            else:
                rois[b, r] = [rng.uniform(80, 90), rng.uniform(50, 60), 0, 2, 2, 2, 0.3]      # no point inside
    pc = np.concatenate(pts).astype(np.float32)
    P = len(pc)
    return {'point_coords': pc, 'point_features': rng.normal(0, 1, (P, C)).astype(np.float32),
            'point_part_offset': rng.uniform(0, 1, (P, 3)).astype(np.float32),
            'point_cls_scores': rng.uniform(0, 1, (P,)).astype(np.float32), 'rois': rois,
            'roi_labels': rng.integers(1, 4, (B, R)).astype(np.int64), 'batch_size': B}


PP_PCR = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]


PP_VOXEL = [0.05, 0.05, 0.1]


PP_KEYPOINTS = 96


PP_LEVELS = (('x_conv1', 1, 16), ('x_conv2', 2, 32), ('x_conv3', 4, 64), ('x_conv4', 8, 64))


def point_path_inputs(seed=51, n_pts=(1500, 1300), c_bev=16, n_roi=12):
    """two ragged frames: ground clutter + 6 object clusters inside x 8..32 m, y -12..12 m; the four sparse feature
    volumes are the occupied voxels of the frame at strides 1/2/4/8 in ascending (b,z,y,x) order with seeded features (what
    VoxelBackBone8x hands to the PFE: .indices / .features only); a seeded BEV map; RoIs on the clusters, next to them and
    far away (empty balls for every grid point); keypoint scores in (0,1)"""
    rng = np.random.default_rng(seed)
    B = len(n_pts)
    pts, rois = [], np.zeros((B, n_roi, 7), np.float32)
    for b, n in enumerate(n_pts):
        ctr = np.stack([rng.uniform(10, 30, 6), rng.uniform(-10, 10, 6), rng.uniform(-1.4, -0.8, 6)], 1)
        k = n // 2
        p = [np.stack([rng.uniform(8, 32, n - k), rng.uniform(-12, 12, n - k), rng.uniform(-2.4, -1.6, n - k)], 1)]
        which = rng.integers(0, 6, k)
        p.append(ctr[which] + rng.normal(0, 1, (k, 3)) * np.array([1.2, 0.6, 0.5]))
        p = np.concatenate(p)
        p[:, 2] = np.clip(p[:, 2], -2.5, 0.5)
        p = p[rng.permutation(n)]
        pts.append(np.concatenate([np.full((n, 1), b), p, rng.uniform(0, 1, (n, 1))], 1))
        for r in range(n_roi):
            c = ctr[r % 6]
            if r < 8:
                rois[b, r] = [c[0] + rng.normal(0, 0.3), c[1] + rng.normal(0, 0.3), c[2], *(np.array([3.9, 1.6, 1.56]) * rng.uniform(0.8, 1.2, 3)),
                              rng.uniform(-np.pi, np.pi)]
            elif r < 10:
                rois[b, r] = [rng.uniform(50, 60), rng.uniform(25, 35), -1, 3.9, 1.6, 1.56, 0.4]     # nothing near
            # last two rows stay all-zero (the padding proposal_layer leaves)
    points = np.concatenate(pts).astype(np.float32)
    lo, vs = np.array(PP_PCR[:3]), np.array(PP_VOXEL)
    levels = {}
    for name, stride, C in PP_LEVELS:
        ijk = np.floor((points[:, 1:4].astype(np.float64) - lo) / (vs * stride)).astype(np.int64)
        c = np.unique(np.concatenate([points[:, :1].astype(np.int64), ijk[:, ::-1]], 1), axis=0)      # sorted (b,z,y,x)
        levels[name] = (c.astype(np.int32), rng.normal(0, 1, (len(c), C)).astype(np.float32))
    bev = rng.normal(0, 1, (B, c_bev, 200, 176)).astype(np.float32)
    scores = rng.uniform(0.05, 1.0, (B * PP_KEYPOINTS,)).astype(np.float32)
    return {'points': points, 'levels': levels, 'bev': bev, 'rois': rois, 'scores': scores, 'batch_size': B}


POST_CFG = {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.1, 'OUTPUT_RAW_SCORE': False, 'EVAL_METRIC': 'kitti',
            'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1, 'NMS_PRE_MAXSIZE': 4096,
                           'NMS_POST_MAXSIZE': 500}}


PV_KEYPOINTS = 256


PV_POINTS = 8000


PV_FIRST_FRAME = 40


PV_KINDS = {'kitti': ('tools/cfgs/active-kitti_models/pv_rcnn_active_crb.yaml', [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1], 4, 8000, 16000),
            'waymo': ('tools/cfgs/active-waymo_models/pv_rcnn_active_crb.yaml', [-75.2, -75.2, -2.0, 75.2, 75.2, 4.0], [0.1, 0.1, 0.15], 5, 24000, 150000)}


PV_GRADS = {'backbone_3d.conv_input.0.weight': np.s_[:], 'backbone_3d.conv3.1.0.weight': np.s_[:16],
            'backbone_2d.blocks.0.1.weight': np.s_[:24], 'pfe.SA_layers.3.mlps.1.0.weight': np.s_[:],
            'roi_head.shared_fc_layer.0.weight': np.s_[:, :384], 'point_head.cls_layers.0.weight': np.s_[:64],
            'dense_head.conv_box.weight': np.s_[:], 'roi_head.roi_grid_pool_layer.mlps.0.0.weight': np.s_[:]}


def pv_grads(kind='kitti'):
    """PV_GRADS for the configuration: the Waymo PFE has two voxel-source SA layers (x_conv3, x_conv4), not four"""
    return {(k.replace('SA_layers.3.', 'SA_layers.1.') if kind == 'waymo' else k): v for k, v in PV_GRADS.items()}


PV_SMALL = ('dense_head.conv_cls.weight', 'dense_head.conv_box.weight', 'dense_head.conv_dir_cls.weight')


def pv_seeded_state(module):
    """seeded_state(module, 71) with the three prediction convolutions of the dense head scaled by 0.05: proposals stay close
    to their anchors (sane box sizes) and the classification loss of random weights stays O(10) instead of O(1000)"""
    sd = seeded_state(module, 71)
    for k in PV_SMALL:
        sd[k] = sd[k] * 0.05
    return sd
