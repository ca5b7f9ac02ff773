"""Shared machinery of the scalar-score baselines (confidence / BALD / MC-dropout variance) and of the embedding
strategies (Coreset, BADGE): one eval pass of the detector over this rank's shard of the pool, per-frame values kept on the
device, ONE all-gather, ONE read-back.

The reference (confidence_sampling.py, bald_sampling.py, montecarlo_sampling.py: identical loops) calls `.item()`-like
Python bookkeeping per frame and ranks only its own sampler shard; here every rank ranks the whole pool like CRB stage 1.
Selection rule kept: `sorted(dict.items(), key=value)` ascending (stable in insertion = pool order), last N entries."""
import numpy as np
import torch

from ..models import load_data_to_gpu
from . import scoring
from .strategy import Strategy


class PoolEvalStrategy(Strategy):
    MC_DROPOUT = False            # enable_dropout() before the pass (bald / montecarlo / badge / crb)

    @staticmethod
    def enable_dropout(model):
        n = 0
        for m in model.modules():
            if m.__class__.__name__.startswith('Dropout'):
                n += 1
                m.train()
        return n

    def _batches(self, ds, frame_indices, batch_size):
        if ds is self.unlabelled_set:
            host = self.iter_pool_batches(frame_indices, batch_size)        # loader workers read ahead of the GPU
        else:
            host = (ds.collate_batch([ds[i] for i in frame_indices[s:s + batch_size]])
                    for s in range(0, len(frame_indices), batch_size))
        for k, batch in enumerate(host):
            chunk = frame_indices[k * batch_size:(k + 1) * batch_size]
            if 'point_frame_offsets' in batch:
                batch['point_frame_counts_host'] = np.diff(batch['point_frame_offsets']).tolist()
            load_data_to_gpu(batch)
            if 'point_frame_offsets' in batch:
                batch['point_frame_offsets'] = batch['point_frame_offsets'].int()
            yield chunk, batch

    @torch.no_grad()
    def eval_pool(self, ds, frame_indices, batch_size, frame_fn, record_points=True):
        """frame_fn(batch_dict, pred_dicts, b) -> tensor row for frame b of the batch; rows are stacked -> (len, D)"""
        model = self.detector
        model.eval()
        if self.MC_DROPOUT:
            self.enable_dropout(model)
        rows, gts = [], []
        for chunk, batch in self._batches(ds, frame_indices, batch_size):
            pred_dicts, _ = model(batch)
            for b in range(len(pred_dicts)):
                if record_points and pred_dicts[b].get('gt_point_stats', None) is not None:
                    gts.append(pred_dicts[b]['gt_point_stats'].reshape(-1))      # device row, gathered in gather_pool
                rows.append(frame_fn(batch, pred_dicts, b).reshape(-1).float())
        self._local_gt_stats = torch.stack(gts, 0) if (record_points and gts) else None
        if not rows:
            return torch.zeros((0, 1), device=next(model.parameters()).device)
        return torch.stack(rows, 0)

    def gather_pool(self, local_rows, n):
        """all-gather this rank's rows of the unlabelled pool; the GT point statistics collected by the same pass are
        gathered with them and recorded for EVERY pool frame on every rank (save_points of the reference's loops)"""
        rank, world = self._world()
        gts = getattr(self, '_local_gt_stats', None)
        widen = gts is not None and gts.shape[0] == local_rows.shape[0]
        if world > 1 or scoring.FORCE_COLLECTIVE:
            # the row width of the collective must be the same on every rank: widen only if EVERY rank collected the
            # statistics of all of its frames (a rank whose frames carry no gt_boxes would otherwise send narrower rows
            # and the all-gather would hang or corrupt)
            import torch.distributed as dist
            flag = torch.tensor([1 if widen else 0], dtype=torch.int32,
                                device=local_rows.device if dist.get_backend() != 'gloo' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            widen = bool(int(flag.item()))
        if widen:
            both = scoring.all_gather_rows(torch.cat([local_rows.float(), gts.float()], 1).contiguous(), n, world)
            self._local_gt_stats = None
            w = local_rows.shape[1]
            self.record_gt_stats(both[:, w:].reshape(n, -1, scoring.GT_STAT_FIELDS))
            return both[:, :w].contiguous()
        return scoring.all_gather_rows(local_rows.contiguous(), n, world)

    def top_n_ascending(self, values, n_select):
        """frame ids of the n_select largest values, in ascending order of value (ties keep pool order) — the tail of the
        reference's sorted dict"""
        order = torch.argsort(values, stable=True)[max(0, values.numel() - n_select):].cpu().tolist()
        return [self.pairs[i][0] for i in order]


class _ScalarScoreSampling(PoolEvalStrategy):
    def frame_value(self, batch, pred_dicts, b):
        raise NotImplementedError

    def query(self, leave_pbar=True, cur_epoch=None):
        if self.cfg.ACTIVE_TRAIN.get('AGGREGATION', 'mean') != 'mean':
            raise NotImplementedError(self.cfg.ACTIVE_TRAIN.AGGREGATION)
        rank, world = self._world()
        n = len(self.pairs)
        mine, _ = scoring.shard_indices(n, rank, world)
        local = self.eval_pool(self.unlabelled_set, mine, self.unlabelled_loader.batch_size or 1, self.frame_value)
        values = self.gather_pool(local, n)[:, 0]
        self.last_values = values
        return self.top_n_ascending(values, self.cfg.ACTIVE_TRAIN.SELECT_NUMS)


def softmax_entropy(logits):
    """-(softmax * log_softmax).sum(1) per box, mean over boxes (NaN for a frame without boxes, like torch.mean of an empty
    tensor in the reference)"""
    lp = torch.log_softmax(logits, dim=1)
    return (-(lp.exp() * lp).sum(dim=1)).mean()
