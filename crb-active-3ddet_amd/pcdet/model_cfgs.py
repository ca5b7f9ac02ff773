"""Model configs as data (the values of tools/cfgs/kitti_models/second.yaml, tools/cfgs/waymo_models/second.yaml and
tools/cfgs/active-kitti_models/pv_rcnn_active_crb.yaml), so the GPU box needs no YAML tree. cfg_from_yaml_file in
pcdet.config reads the reference's own YAMLs when they are available."""
from .config import EasyDict


def _anchor(cls, size, bottom, matched, unmatched):
    return {'class_name': cls, 'anchor_sizes': [size], 'anchor_rotations': [0, 1.57],
            'anchor_bottom_heights': [bottom], 'align_center': False, 'feature_map_stride': 8,
            'matched_threshold': matched, 'unmatched_threshold': unmatched}


def _dense_head(anchors, predict_when_training_cfg=None):
    return {
        'NAME': 'AnchorHeadSingle', 'CLASS_AGNOSTIC': False, 'USE_DIRECTION_CLASSIFIER': True,
        'DIR_OFFSET': 0.78539, 'DIR_LIMIT_OFFSET': 0.0, 'NUM_DIR_BINS': 2,
        'ANCHOR_GENERATOR_CONFIG': anchors,
        'TARGET_ASSIGNER_CONFIG': {'NAME': 'AxisAlignedTargetAssigner', 'POS_FRACTION': -1.0, 'SAMPLE_SIZE': 512,
                                   'NORM_BY_NUM_EXAMPLES': False, 'MATCH_HEIGHT': False, 'BOX_CODER': 'ResidualCoder'},
        'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0, 'loc_weight': 2.0, 'dir_weight': 0.2,
                                         'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}},
    }


KITTI_ANCHORS = [_anchor('Car', [3.9, 1.6, 1.56], -1.78, 0.6, 0.45),
                 _anchor('Pedestrian', [0.8, 0.6, 1.73], -0.6, 0.5, 0.35),
                 _anchor('Cyclist', [1.76, 0.6, 1.73], -0.6, 0.5, 0.35)]
WAYMO_ANCHORS = [_anchor('Vehicle', [4.7, 2.1, 1.7], 0, 0.55, 0.4),
                 _anchor('Pedestrian', [0.91, 0.86, 1.73], 0, 0.5, 0.35),
                 _anchor('Cyclist', [1.78, 0.84, 1.78], 0, 0.5, 0.35)]

_BEV = {'NAME': 'BaseBEVBackbone', 'LAYER_NUMS': [5, 5], 'LAYER_STRIDES': [1, 2], 'NUM_FILTERS': [128, 256],
        'UPSAMPLE_STRIDES': [1, 2], 'NUM_UPSAMPLE_FILTERS': [256, 256]}


def second_cfg(kind='kitti'):
    anchors = KITTI_ANCHORS if kind == 'kitti' else WAYMO_ANCHORS
    names = ['Car', 'Pedestrian', 'Cyclist'] if kind == 'kitti' else ['Vehicle', 'Pedestrian', 'Cyclist']
    nms_thresh = 0.01 if kind == 'kitti' else 0.7
    return EasyDict({
        'CLASS_NAMES': names,
        'MODEL': {
            'NAME': 'SECONDNet',
            'VFE': {'NAME': 'MeanVFE'},
            'BACKBONE_3D': {'NAME': 'VoxelBackBone8x'},
            'MAP_TO_BEV': {'NAME': 'HeightCompression', 'NUM_BEV_FEATURES': 256},
            'BACKBONE_2D': dict(_BEV),
            'DENSE_HEAD': _dense_head(anchors),
            'POST_PROCESSING': {
                'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.1, 'OUTPUT_RAW_SCORE': False,
                'EVAL_METRIC': 'kitti',
                'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': nms_thresh,
                               'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
        },
        'OPTIMIZATION': {'BATCH_SIZE_PER_GPU': 4, 'NUM_EPOCHS': 80, 'OPTIMIZER': 'adam_onecycle', 'LR': 0.003,
                         'WEIGHT_DECAY': 0.01, 'MOMENTUM': 0.9, 'MOMS': [0.95, 0.85], 'PCT_START': 0.4,
                         'DIV_FACTOR': 10, 'GRAD_NORM_CLIP': 10},
    })


def pv_rcnn_cfg(kind='kitti'):
    """values of tools/cfgs/active-kitti_models/pv_rcnn_active_crb.yaml; kind='waymo': the differences of
    tools/cfgs/active-waymo_models/pv_rcnn_active_crb.yaml applied on top (anchors, 4096 keypoints, bev/x_conv3/x_conv4/
    raw_points feature sources, NMS sizes and thresholds, K1=3 / K2=2 / SELECT_NUMS=400)"""
    assert kind in ('kitti', 'waymo')
    if kind == 'waymo':
        c = pv_rcnn_cfg('kitti')
        c.CLASS_NAMES = ['Vehicle', 'Pedestrian', 'Cyclist']
        c.DATA_CONFIG.DATASET = 'WaymoDataset'
        m = c.MODEL
        m.DENSE_HEAD = EasyDict(_dense_head(WAYMO_ANCHORS))
        m.PFE.NUM_KEYPOINTS = 4096
        m.PFE.FEATURES_SOURCE = ['bev', 'x_conv3', 'x_conv4', 'raw_points']
        m.POINT_HEAD.NUM_KEYPOINTS = 4096
        m.ROI_HEAD.NMS_CONFIG.TEST.NMS_PRE_MAXSIZE = 4096
        m.ROI_HEAD.NMS_CONFIG.TEST.NMS_THRESH = 0.85
        m.POST_PROCESSING.EVAL_METRIC = 'waymo'
        m.POST_PROCESSING.NMS_CONFIG.NMS_THRESH = 0.7
        c.OPTIMIZATION.WEIGHT_DECAY = 0.001
        c.ACTIVE_TRAIN.update({'PRE_TRAIN_SAMPLE_NUMS': 400, 'SELECT_NUMS': 400, 'TOTAL_BUDGET_NUMS': 2000})
        c.ACTIVE_TRAIN.ACTIVE_CONFIG.update({'K1': 3, 'K2': 2})
        return c
    sa = lambda f, mlp, r, ns: {'DOWNSAMPLE_FACTOR': f, 'MLPS': [list(mlp), list(mlp)], 'POOL_RADIUS': list(r),
                                'NSAMPLE': list(ns)}
    return EasyDict({
        'CLASS_NAMES': ['Car', 'Pedestrian', 'Cyclist'],
        'DATA_CONFIG': {'DATASET': 'KittiDataset'},
        'MODEL': {
            'NAME': 'PVRCNN',
            'VFE': {'NAME': 'MeanVFE'},
            'BACKBONE_3D': {'NAME': 'VoxelBackBone8x'},
            'MAP_TO_BEV': {'NAME': 'HeightCompression', 'NUM_BEV_FEATURES': 256},
            'BACKBONE_2D': dict(_BEV),
            'DENSE_HEAD': _dense_head(KITTI_ANCHORS),
            'PFE': {
                'NAME': 'VoxelSetAbstraction', 'POINT_SOURCE': 'raw_points', 'NUM_KEYPOINTS': 2048,
                'NUM_OUTPUT_FEATURES': 128, 'SAMPLE_METHOD': 'FPS',
                'FEATURES_SOURCE': ['bev', 'x_conv1', 'x_conv2', 'x_conv3', 'x_conv4', 'raw_points'],
                'SA_LAYER': {
                    'raw_points': {'MLPS': [[16, 16], [16, 16]], 'POOL_RADIUS': [0.4, 0.8], 'NSAMPLE': [16, 16]},
                    'x_conv1': sa(1, (16, 16), (0.4, 0.8), (16, 16)),
                    'x_conv2': sa(2, (32, 32), (0.8, 1.2), (16, 32)),
                    'x_conv3': sa(4, (64, 64), (1.2, 2.4), (16, 32)),
                    'x_conv4': sa(8, (64, 64), (2.4, 4.8), (16, 32)),
                }},
            'POINT_HEAD': {
                'NAME': 'PointHeadSimple', 'CLS_FC': [256, 256], 'CLASS_AGNOSTIC': True,
                'USE_POINT_FEATURES_BEFORE_FUSION': True, 'NUM_KEYPOINTS': 2048,
                'TARGET_CONFIG': {'GT_EXTRA_WIDTH': [0.2, 0.2, 0.2]},
                'LOSS_CONFIG': {'LOSS_REG': 'smooth-l1', 'LOSS_WEIGHTS': {'point_cls_weight': 1.0}}},
            'ROI_HEAD': {
                'NAME': 'PVRCNNHead', 'CLASS_AGNOSTIC': True, 'SAMPLING_ROUND': 5, 'SHARED_FC': [256, 256],
                'CLS_FC': [256, 256], 'REG_FC': [256, 256], 'DP_RATIO': 0.3,
                'NMS_CONFIG': {
                    'TRAIN': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 9000,
                              'NMS_POST_MAXSIZE': 512, 'NMS_THRESH': 0.8},
                    'TEST': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 1024,
                             'NMS_POST_MAXSIZE': 128, 'NMS_THRESH': 0.7}},
                'ROI_GRID_POOL': {'GRID_SIZE': 6, 'MLPS': [[64, 64], [64, 64]], 'POOL_RADIUS': [0.8, 1.6],
                                  'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool'},
                'TARGET_CONFIG': {'BOX_CODER': 'ResidualCoder', 'ROI_PER_IMAGE': 128, 'FG_RATIO': 0.5,
                                  'SAMPLE_ROI_BY_EACH_CLASS': True, 'CLS_SCORE_TYPE': 'roi_iou', 'CLS_FG_THRESH': 0.75,
                                  'CLS_BG_THRESH': 0.25, 'CLS_BG_THRESH_LO': 0.1, 'HARD_BG_RATIO': 0.8,
                                  'REG_FG_THRESH': 0.55},
                'LOSS_CONFIG': {'CLS_LOSS': 'BinaryCrossEntropy', 'REG_LOSS': 'smooth-l1',
                                'CORNER_LOSS_REGULARIZATION': True,
                                'LOSS_WEIGHTS': {'rcnn_cls_weight': 1.0, 'rcnn_reg_weight': 1.0,
                                                 'rcnn_corner_weight': 1.0,
                                                 'code_weights': [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]}}},
            'POST_PROCESSING': {
                'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.1, 'OUTPUT_RAW_SCORE': False,
                'EVAL_METRIC': 'kitti',
                'NMS_CONFIG': {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.1,
                               'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
        },
        'OPTIMIZATION': {'OPTIMIZER': 'adam_onecycle', 'LR': 0.01, 'WEIGHT_DECAY': 0.01, 'MOMENTUM': 0.9,
                         'MOMS': [0.95, 0.85], 'PCT_START': 0.4, 'DIV_FACTOR': 10, 'GRAD_NORM_CLIP': 10},
        'ACTIVE_TRAIN': {'METHOD': 'crb', 'AGGREGATION': 'mean', 'PRE_TRAIN_SAMPLE_NUMS': 100,
                         'PRE_TRAIN_EPOCH_NUMS': 40, 'TRAIN_RESUME': True, 'SELECT_NUMS': 100,
                         'SELECT_LABEL_EPOCH_INTERVAL': 40, 'TOTAL_BUDGET_NUMS': 600,
                         'ACTIVE_CONFIG': {'K1': 5, 'K2': 3, 'BANDWIDTH': 5, 'CLUSTERING': 'kmeans++'}},
    })
