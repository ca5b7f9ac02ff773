"""class_agnostic_nms / multi_classes_nms with the reference's call contract (pcdet/models/model_utils/model_nms_utils.py:6-66).

Both reduce to one step — keep the boxes passing the score threshold, take the NMS_PRE_MAXSIZE best, run the configured
rotated NMS of iou3d_nms_utils on them, keep NMS_POST_MAXSIZE — written once here (`_survivors`)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def _survivors(scores, boxes, nms_config, score_thresh):
    """indices INTO `scores` of the boxes that survive threshold + top-k + NMS, best first (LongTensor, possibly empty)"""
    cand = torch.arange(scores.shape[0], device=scores.device)
    if score_thresh is not None:
        cand = cand[scores >= score_thresh]
    if cand.numel() == 0:
        return cand
    k = min(int(nms_config.NMS_PRE_MAXSIZE), int(cand.numel()))
    top_scores, order = torch.topk(scores[cand], k=k)
    cand = cand[order]
    nms_fn = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)
    keep, _ = nms_fn(boxes[cand][:, 0:7], top_scores, nms_config.NMS_THRESH, **nms_config)
    return cand[keep[:nms_config.NMS_POST_MAXSIZE]]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """-> (selected indices into the inputs, their scores)"""
    sel = _survivors(box_scores, box_preds, nms_config, score_thresh)
    return sel, box_scores[sel]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """per-class NMS on (N, num_class) scores -> (scores, labels, boxes) concatenated over the classes in class order"""
    out_scores, out_labels, out_boxes = [], [], []
    for cls in range(cls_scores.shape[1]):
        col = cls_scores[:, cls]
        sel = _survivors(col, box_preds, nms_config, score_thresh)
        out_scores.append(col[sel])
        out_labels.append(torch.full((sel.numel(),), cls, dtype=torch.long, device=col.device))
        out_boxes.append(box_preds[sel])
    return torch.cat(out_scores, dim=0), torch.cat(out_labels, dim=0), torch.cat(out_boxes, dim=0)
