"""Box geometry used on the hot path (pcdet/utils/box_utils.py:28-80,240-298), batched where the reference loops."""
import numpy as np
import torch

from . import common_utils


_CONSTANTS = {}


def _constant(like, key, values):
    """small constant tensor on like's device / dtype, uploaded once (a new_tensor(list) per call is a pageable host-to-device
    copy = a host synchronisation in the middle of a training step)"""
    k = (key, like.device, like.dtype)
    t = _CONSTANTS.get(k)
    if t is None:
        t = _CONSTANTS[k] = like.new_tensor(values)
    return t


def boxes_to_corners_3d(boxes3d):
    """(N,7) -> (N,8,3) corner order of box_utils.py:58-80"""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    template = _constant(boxes3d, 'corner_template', (
        [0.5, 0.5, -0.5], [0.5, -0.5, -0.5], [-0.5, -0.5, -0.5], [-0.5, 0.5, -0.5],
        [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [-0.5, -0.5, 0.5], [-0.5, 0.5, 0.5],
    ))
    corners3d = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners3d = common_utils.rotate_points_along_z(corners3d.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    corners3d += boxes3d[:, None, 0:3]
    return corners3d.numpy() if is_numpy else corners3d


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large = boxes3d.clone()
    large[:, 3:6] += _constant(boxes3d, ('extra_width',) + tuple(float(w) for w in extra_width), extra_width)[None, :]
    return large


def boxes_iou_normal(boxes_a, boxes_b):
    """axis-aligned 2-D IoU, boxes (..,N,4)/(..,M,4) [x1,y1,x2,y2] -> (..,N,M) (box_utils.py:246-269)"""
    assert boxes_a.shape[-1] == boxes_b.shape[-1] == 4
    ax1, ay1, ax2, ay2 = [boxes_a[..., :, None, i] for i in range(4)]
    bx1, by1, bx2, by2 = [boxes_b[..., None, :, i] for i in range(4)]
    x_len = torch.clamp_min(torch.min(ax2, bx2) - torch.max(ax1, bx1), min=0)
    y_len = torch.clamp_min(torch.min(ay2, by2) - torch.max(ay1, by1), min=0)
    area_a = (ax2 - ax1) * (ay2 - ay1)
    area_b = (bx2 - bx1) * (by2 - by1)
    inter = x_len * y_len
    return inter / torch.clamp_min(area_a + area_b - inter, min=1e-6)


def boxes3d_lidar_to_aligned_bev_boxes(boxes3d):
    """(..,7) -> (..,4): heading snapped to the nearest axis (box_utils.py:272-283)"""
    rot = common_utils.limit_period(boxes3d[..., 6], offset=0.5, period=np.pi).abs()
    dims = torch.where(rot[..., None] < np.pi / 4, boxes3d[..., [3, 4]], boxes3d[..., [4, 3]])
    return torch.cat((boxes3d[..., 0:2] - dims / 2, boxes3d[..., 0:2] + dims / 2), dim=-1)


def boxes3d_nearest_bev_iou(boxes_a, boxes_b):
    """(..,N,7),(..,M,7) -> (..,N,M) (box_utils.py:286-298)"""
    return boxes_iou_normal(boxes3d_lidar_to_aligned_bev_boxes(boxes_a), boxes3d_lidar_to_aligned_bev_boxes(boxes_b))


def mask_boxes_outside_range_numpy(boxes, limit_range, min_num_corners=1):
    """boxes (N,7+) numpy -> (N) bool: boxes with at least min_num_corners corners inside limit_range (box_utils.py:56-72)"""
    import numpy as np
    if boxes.shape[1] > 7:
        boxes = boxes[:, 0:7]
    if boxes.shape[0] == 0:
        return np.zeros((0,), dtype=bool)
    corners = boxes_to_corners_3d(boxes)
    corners = corners.numpy() if hasattr(corners, 'numpy') else np.asarray(corners)
    lr = np.asarray(limit_range, dtype=np.float32)
    mask = ((corners >= lr[0:3]) & (corners <= lr[3:6])).all(axis=2)
    return mask.sum(axis=1) >= min_num_corners
