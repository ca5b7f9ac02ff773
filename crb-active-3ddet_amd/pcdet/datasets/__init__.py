"""Dataset side of the hot path: the synthetic KITTI/Waymo-shaped dataset used for benchmarking and tests, the
collate layout of DatasetTemplate.collate_batch (pcdet/datasets/dataset.py:160-229), the rank-strided eval sampler of
pcdet/datasets/__init__.py:26-46 and the two loader builders the reference's tools/train.py and active loop call
(build_dataloader :49-78, build_active_dataloader :80-181) with the reference's signatures and return tuples.

Real-dataset readers / augmentors are out of scope: the registry below answers 'KittiDataset' and
'WaymoDataset' with synthetic clouds of that shape — ONLY when the config asks for it (dataset_cfg.SYNTHETIC present, or
CRB_SYNTHETIC_DATA=1 in the environment). A reference config that names a real dataset (DATA_PATH / INFO_PATH / root_path)
without that opt-in is refused loudly instead of training and selecting on fake frames."""
import os
import random
import warnings

import torch
from torch.utils.data import DataLoader

from ..config import cfg
from ..utils import common_utils
from .sampler import DistributedSampler  # noqa: F401
from .synthetic_dataset import SyntheticDataset, build_synthetic_dataloader  # noqa: F401


class _ConfiguredSynthetic(SyntheticDataset):
    """SyntheticDataset behind the reference's dataset constructor (dataset_cfg, class_names, root_path, training, logger).
    Optional dataset_cfg.SYNTHETIC = {NUM_FRAMES, N_POINTS, FIRST_FRAME, DEVICE_VOXELIZE}."""
    KIND = 'kitti'

    def __init__(self, dataset_cfg=None, class_names=None, training=True, root_path=None, logger=None):
        has_key = dataset_cfg is not None and dataset_cfg.get('SYNTHETIC', None) is not None
        if not has_key and os.environ.get('CRB_SYNTHETIC_DATA', '0') != '1':
            raise NotImplementedError(
                "%s: this build has no reader for the real dataset (out of scope: SURVEY §2.1). The name is served with "
                "SYNTHETIC %s-shaped clouds only on request: put SYNTHETIC: {NUM_FRAMES: .., N_POINTS: ..} into DATA_CONFIG "
                "(or export CRB_SYNTHETIC_DATA=1); DATA_PATH / INFO_PATH / root_path (%r) are not read."
                % (type(self).__name__, self.KIND, root_path))
        msg = '%s serves SYNTHETIC %s-shaped frames (DATA_PATH / INFO_PATH / root_path are ignored)' % (type(self).__name__, self.KIND)
        if logger is not None:
            logger.warning(msg)
        elif not has_key:
            warnings.warn(msg)
        syn = dict((dataset_cfg or {}).get('SYNTHETIC', {}) or {})
        super().__init__(num_frames=int(syn.get('NUM_FRAMES', 64)),
                         n_points=int(syn.get('N_POINTS', 20000 if self.KIND == 'kitti' else 160000)),
                         kind=self.KIND, training=training, first_frame=int(syn.get('FIRST_FRAME', 0)),
                         device_voxelize=bool(syn.get('DEVICE_VOXELIZE', True)), class_names=class_names)
        self.dataset_cfg, self.root_path, self.logger = dataset_cfg, root_path, logger

    def __getstate__(self):
        d = dict(self.__dict__)
        d.pop('logger', None)
        d['_voxel_generator'] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.logger = None


class KittiDataset(_ConfiguredSynthetic):
    KIND = 'kitti'


class WaymoDataset(_ConfiguredSynthetic):
    KIND = 'waymo'


__all__ = {
    'KittiDataset': KittiDataset,
    'WaymoDataset': WaymoDataset,
    'SyntheticDataset': SyntheticDataset,
}


def _samplers(dist, training, *datasets):
    if not dist:
        return [None] * len(datasets)
    if training:
        return [torch.utils.data.distributed.DistributedSampler(d) for d in datasets]
    rank, world_size = common_utils.get_dist_info()
    return [DistributedSampler(d, world_size, rank, shuffle=False) for d in datasets]


class LookaheadLoader:
    """A DataLoader whose batches know their successor. Iterating yields the loader's own batch dicts, one
    step behind: every batch carries `batch['_crb_next']` = the NEXT batch dict (None for the last), which the loader workers have
    produced by then anyway. `Detector3DTemplate.forward` takes it as the cue to enqueue the next batch's voxel generator, table
    marks and keypoint sampling behind this batch's dense half (`prefetch_sparse`): the reference's unmodified `train_one_epoch`
    (`batch = next(dataloader_iter); model_func(model, batch)`, tools/train_utils/train_utils.py:26-44) then runs without the
    per-step read-back, like a caller that pipelines by hand. Everything else (len, dataset, batch_size, num_workers, sampler, ...)
    is the wrapped loader's. CRB_LOADER_LOOKAHEAD=0 hands out plain DataLoaders."""

    def __init__(self, loader):
        self.__dict__['_loader'] = loader

    def __getattr__(self, name):
        return getattr(self.__dict__['_loader'], name)

    def __setattr__(self, name, value):
        setattr(self.__dict__['_loader'], name, value)

    def __len__(self):
        return len(self.__dict__['_loader'])

    def __iter__(self):
        it = iter(self.__dict__['_loader'])
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            if isinstance(cur, dict):
                cur['_crb_next'] = nxt if isinstance(nxt, dict) else None
            yield cur
            cur = nxt
        if isinstance(cur, dict):
            cur['_crb_next'] = None
        yield cur


LOOKAHEAD = __import__('os').environ.get('CRB_LOADER_LOOKAHEAD', '1') == '1'


def _loader(dataset, batch_size, workers, sampler, training):
    dl = DataLoader(dataset, batch_size=batch_size, pin_memory=True, num_workers=workers,
                    shuffle=(sampler is None) and training, collate_fn=dataset.collate_batch, drop_last=False,
                    sampler=sampler, timeout=0)
    return LookaheadLoader(dl) if (LOOKAHEAD and training) else dl


def build_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=4, logger=None, training=True,
                     merge_all_iters_to_one_epoch=False, total_epochs=0):
    """-> dataset, dataloader, sampler (pcdet/datasets/__init__.py:49-78)"""
    dataset = __all__[dataset_cfg.DATASET](dataset_cfg=dataset_cfg, class_names=class_names, root_path=root_path,
                                           training=training, logger=logger)
    if merge_all_iters_to_one_epoch:
        assert hasattr(dataset, 'merge_all_iters_to_one_epoch')
        dataset.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
    sampler, = _samplers(dist, training, dataset)
    return dataset, _loader(dataset, batch_size, workers, sampler, training), sampler


def build_active_dataloader(dataset_cfg, class_names, batch_size, dist, root_path=None, workers=4, logger=None,
                            training=True, merge_all_iters_to_one_epoch=False, total_epochs=0, active_training=None):
    """-> labelled_set, unlabelled_set, dataloader_labelled, dataloader_unlabelled, sampler_labelled, sampler_unlabelled
    (pcdet/datasets/__init__.py:80-181). active_training = [selected ids, selected infos, unselected ids, unselected infos]
    rebuilds the split after a selection round; None draws the initial random split of
    cfg.ACTIVE_TRAIN.PRE_TRAIN_SAMPLE_NUMS labelled frames with the global `random` state, like the reference."""
    make = lambda tr: __all__[dataset_cfg.DATASET](dataset_cfg=dataset_cfg, class_names=class_names, root_path=root_path,
                                                   training=tr, logger=logger)
    dataset, labelled_set, unlabelled_set = make(training), make(True), make(False)
    waymo = cfg.DATA_CONFIG.DATASET == 'WaymoDataset'
    if active_training is not None:
        if waymo:
            labelled_set.frame_ids, labelled_set.infos = active_training[0], active_training[1]
            unlabelled_set.frame_ids, unlabelled_set.infos = active_training[2], active_training[3]
        else:
            labelled_set.sample_id_list, labelled_set.kitti_infos = active_training[0], active_training[1]
            unlabelled_set.sample_id_list, unlabelled_set.kitti_infos = active_training[2], active_training[3]
    else:
        k = cfg.ACTIVE_TRAIN.PRE_TRAIN_SAMPLE_NUMS
        if waymo:
            infos = list(dataset.infos)
            random.shuffle(infos)
            labelled_set.infos, unlabelled_set.infos = infos[:k], infos[k:]
            labelled_set.frame_ids = [i['frame_id'] for i in labelled_set.infos]
            unlabelled_set.frame_ids = [i['frame_id'] for i in unlabelled_set.infos]
        else:
            pairs = list(zip(dataset.sample_id_list, dataset.kitti_infos))
            random.shuffle(pairs)
            labelled_set.sample_id_list, labelled_set.kitti_infos = zip(*pairs[:k])
            unlabelled_set.sample_id_list, unlabelled_set.kitti_infos = zip(*pairs[k:])
    for s in (labelled_set, unlabelled_set):
        s.sync_id_views(waymo)
    if merge_all_iters_to_one_epoch:
        assert hasattr(dataset, 'merge_all_iters_to_one_epoch')
        labelled_set.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
        unlabelled_set.merge_all_iters_to_one_epoch(merge=True, epochs=total_epochs)
    sampler_labelled, sampler_unlabelled = _samplers(dist, training, labelled_set, unlabelled_set)
    dataloader_labelled = _loader(labelled_set, batch_size, workers, sampler_labelled, training)
    dataloader_unlabelled = _loader(unlabelled_set, batch_size, workers, sampler_unlabelled, training)
    del dataset
    return labelled_set, unlabelled_set, dataloader_labelled, dataloader_unlabelled, sampler_labelled, sampler_unlabelled
