"""EntropySampling: rank the pool by the stage-1 label entropy only (the ENTROPY baseline of the reference,
pcdet/query_strategies/entropy_sampling.py), on the same device-resident records as CRB."""
import torch

from . import scoring
from .crb_sampling import CRBSampling


class EntropySampling(CRBSampling):
    def query(self, leave_pbar=True, cur_epoch=None):
        rank, world = self._world()
        n = len(self.pairs)
        mine, _ = scoring.shard_indices(n, rank, world)
        records = scoring.all_gather_rows(self.score_pool(mine, self.unlabelled_loader.batch_size or 1), n, world)
        order = torch.argsort(records[:, 0], stable=True).flip(0)[:self.cfg.ACTIVE_TRAIN.SELECT_NUMS].cpu().tolist()
        return [self.pairs[i][0] for i in order]
