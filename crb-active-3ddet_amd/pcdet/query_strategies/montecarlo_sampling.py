"""MonteCarloSampling (pcdet/query_strategies/montecarlo_sampling.py:18-80): variance over the MC-dropout passes of
sigmoid(rcnn_cls) plus variance of rcnn_reg, averaged over the frame's RoIs.

The reference reads `pred_dicts[0]['rcnn_cls']` / `['rcnn_reg']` (`:52-53`), keys its own post_processing no longer emits
(`detector3d_template.py:390-406` writes `batch_rcnn_cls` / `batch_rcnn_reg`, the per-frame slices) — as shipped it
raises KeyError. This implementation uses the per-frame slices, the evident intent."""
import torch

from .pool_eval import _ScalarScoreSampling


class MonteCarloSampling(_ScalarScoreSampling):
    MC_DROPOUT = True

    def frame_value(self, batch, pred_dicts, b):
        d = pred_dicts[b]
        return torch.var(torch.sigmoid(d['batch_rcnn_cls']), 0).mean() + torch.var(d['batch_rcnn_reg'], 0).mean()
