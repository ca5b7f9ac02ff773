"""seeded rotated-box sets for IoU / NMS tests: clustered detections around a few objects (many overlaps) with
margin-aware construction (pairs whose IoU sits within 1e-4 of a threshold can be filtered by the caller)"""
import numpy as np


def detection_boxes(rng, n, n_obj=40, spread=0.6, xy=((0, 70), (-40, 40))):
    sizes = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]], np.float32)
    objs = np.stack([rng.uniform(*xy[0], n_obj), rng.uniform(*xy[1], n_obj), rng.uniform(-1.5, -0.5, n_obj),
                     rng.integers(0, 3, n_obj), rng.uniform(-np.pi, np.pi, n_obj)], axis=1)
    k = rng.integers(0, n_obj, n)
    o = objs[k]
    s = sizes[o[:, 3].astype(int)] * rng.uniform(0.8, 1.2, (n, 3))
    boxes = np.concatenate([o[:, 0:2] + rng.normal(0, spread, (n, 2)), o[:, 2:3] + rng.normal(0, 0.1, (n, 1)), s,
                            (o[:, 4] + rng.normal(0, 0.2, n))[:, None]], axis=1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    return boxes, scores
