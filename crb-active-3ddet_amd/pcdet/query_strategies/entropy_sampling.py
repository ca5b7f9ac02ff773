"""EntropySampling (pcdet/query_strategies/entropy_sampling.py:7-69): mean softmax entropy of 'pred_logits' (the full class
scores of the final boxes), dropout OFF (the reference only calls model.eval()); the SELECT_NUMS frames with the largest
value, in ascending order of value."""
from .pool_eval import _ScalarScoreSampling, softmax_entropy


class EntropySampling(_ScalarScoreSampling):
    MC_DROPOUT = False

    def frame_value(self, batch, pred_dicts, b):
        return softmax_entropy(pred_dicts[b]['pred_logits'])
