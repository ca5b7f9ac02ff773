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
