"""BALDSampling (pcdet/query_strategies/bald_sampling.py:8-70): mean softmax entropy of 'pred_logits' of the final
boxes with MC dropout enabled."""
from .pool_eval import _ScalarScoreSampling, softmax_entropy


class BALDSampling(_ScalarScoreSampling):
    MC_DROPOUT = True

    def frame_value(self, batch, pred_dicts, b):
        return softmax_entropy(pred_dicts[b]['pred_logits'])
