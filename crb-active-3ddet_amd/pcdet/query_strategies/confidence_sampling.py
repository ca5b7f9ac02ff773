"""ConfidenceSampling (pcdet/query_strategies/confidence_sampling.py:12-69): mean softmax entropy of the RPN class
scores of the final boxes ('confidence' record), dropout OFF."""
from .pool_eval import _ScalarScoreSampling, softmax_entropy


class ConfidenceSampling(_ScalarScoreSampling):
    MC_DROPOUT = False

    def frame_value(self, batch, pred_dicts, b):
        return softmax_entropy(pred_dicts[b]['confidence'])
