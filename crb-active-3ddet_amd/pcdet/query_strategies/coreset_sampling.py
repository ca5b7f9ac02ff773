"""CoresetSampling (pcdet/query_strategies/coreset_sampling.py:7-132): k-centre greedy on the RoI-head shared features
(128 RoIs x 256 = 32768-d per frame) of the unlabelled pool against the labelled set.

`furthest_first` keeps the reference's arithmetic — including its initialisation with the MEAN (not the min) squared
distance to the labelled embeddings (`:38`) — but the `for j in range(m): min_dist[j] = min(...)` host loop (`:47-48`,
O(n*m) Python iterations with a device sync each) is one `torch.minimum`."""
import torch

from . import scoring
from .pool_eval import PoolEvalStrategy


def pairwise_squared_distances(x, y):
    n, m = x.shape[0], y.shape[0]
    x, y = x.reshape(n, -1), y.reshape(m, -1)
    dist = (x ** 2).sum(1).view(n, 1) + (y ** 2).sum(1).view(1, m) - 2.0 * torch.mm(x, y.t().contiguous())
    dist = torch.where(dist != dist, torch.zeros_like(dist), dist)
    return torch.clamp(dist, 0.0, float('inf'))


def furthest_first(X, X_set, n):
    """-> list of n row indices of X"""
    m = X.shape[0]
    X, X_set = X.reshape(m, -1), X_set.reshape(X_set.shape[0], -1)
    min_dist = pairwise_squared_distances(X, X_set).mean(1)
    idxs = []
    for i in range(n):
        idx = torch.argmax(min_dist)
        idxs.append(idx)
        if i < n - 1:
            min_dist = torch.minimum(min_dist, pairwise_squared_distances(X, X[idx].unsqueeze(0))[:, 0])
    return [int(v) for v in torch.stack(idxs).cpu().tolist()] if idxs else []


class CoresetSampling(PoolEvalStrategy):
    MC_DROPOUT = False

    def _embedding(self, batch, pred_dicts, b):
        # the reference takes pred_dicts[0]['embeddings'] (the batch's shared features, B*128 x 256) and views it as
        # (-1, 128, 256): one row per frame; frame b of the batch is row b
        width = self.cfg.MODEL.ROI_HEAD.SHARED_FC[-1]
        return pred_dicts[0]['embeddings'].reshape(-1, 128, width)[b]

    def query(self, leave_pbar=True, cur_epoch=None):
        rank, world = self._world()
        n = len(self.pairs)
        bs = self.unlabelled_loader.batch_size or 1
        mine, _ = scoring.shard_indices(n, rank, world)
        unl = self.gather_pool(self.eval_pool(self.unlabelled_set, mine, bs, self._embedding), n)
        nl = len(self.labelled_set)
        lmine, _ = scoring.shard_indices(nl, rank, world)
        lab = scoring.all_gather_rows(self.eval_pool(self.labelled_set, lmine, self.labelled_loader.batch_size or 1,
                                                     self._embedding, record_points=False).contiguous(), nl, world)
        sel = furthest_first(unl, lab, self.cfg.ACTIVE_TRAIN.SELECT_NUMS)
        return [self.pairs[i][0] for i in sel]
