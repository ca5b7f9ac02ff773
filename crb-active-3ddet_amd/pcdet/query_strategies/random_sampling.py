"""RandomSampling (pcdet/query_strategies/random_sampling.py:8-59). The reference runs the detector over the whole pool only
to call save_points() on every frame (the GT point statistics its caller pickles), then shuffles the frame ids with the
global `random` state. The statistics depend on the points and gt boxes alone, so here they come from ONE device pass per
batch without the detector (crb_gt_point_stats), sharded over ranks and all-gathered like the scoring strategies."""
import random

from . import scoring
from .strategy import Strategy


class RandomSampling(Strategy):
    def query(self, leave_pbar=True, cur_epoch=None):
        rank, world = self._world()
        n = len(self.pairs)
        all_frames = [p[0] for p in self.pairs]
        if len(self.bbox_records) == 0:
            mine, _ = scoring.shard_indices(n, rank, world)
            local = self.gt_stats_pool(mine, self.unlabelled_loader.batch_size or 1)
            stats = scoring.all_gather_rows(local.reshape(local.shape[0], -1).contiguous(), n, world)
            self.record_gt_stats(stats.reshape(n, -1, scoring.GT_STAT_FIELDS), all_frames)
        random.shuffle(all_frames)
        return all_frames[:self.cfg.ACTIVE_TRAIN.SELECT_NUMS]
