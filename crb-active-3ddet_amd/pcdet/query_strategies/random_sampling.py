import random

from .strategy import Strategy


class RandomSampling(Strategy):
    def query(self, leave_pbar=True, cur_epoch=None):
        ids = [p[0] for p in self.pairs]
        random.shuffle(ids)
        return ids[:self.cfg.ACTIVE_TRAIN.SELECT_NUMS]
