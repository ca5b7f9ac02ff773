"""Rank-strided sampler with wrap-around padding — the semantics of the reference's custom DistributedSampler
(pcdet/datasets/__init__.py:26-46): rank r of W takes indices r, r+W, ... of the (optionally shuffled) padded list."""
import math

import torch
from torch.utils.data import Sampler


class DistributedSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            ok = dist.is_available() and dist.is_initialized()
            num_replicas = num_replicas if num_replicas is not None else (dist.get_world_size() if ok else 1)
            rank = rank if rank is not None else (dist.get_rank() if ok else 0)
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) * 1.0 / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(len(self.dataset), generator=g).tolist()
        else:
            indices = torch.arange(len(self.dataset)).tolist()
        indices += indices[:(self.total_size - len(indices))]
        assert len(indices) == self.total_size
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples
