"""select_active_labels — the call the active training loop makes every SELECT_LABEL_EPOCH_INTERVAL epochs
(pcdet/utils/active_training_utils.py:240-325; caller tools/train_utils/train_active_utils.py:301-314): build the query
strategy, query(), save_active_labels(), move the selected frames from the unlabelled to the labelled split and rebuild
both loaders. Same signature, same files written, same return value.

The pseudo-label helpers of the same reference file (save_active_label_epoch and friends, :20-237) belong to the ST3D
self-training leftovers that the active loop never calls (SURVEY §2.3 "dead path") and are not provided."""
import os
import pickle as pkl

from .. import query_strategies
from ..config import cfg
from ..datasets import build_active_dataloader


def move_selected_frames(strategy, selected_frames):
    """-> [selected ids, selected infos, unselected ids, unselected infos] as tuples: the labelled split followed by the
    selected pool frames in POOL order, the pool without them (active_training_utils.py:276-300). The reference walks
    strategy.pairs and calls list.remove() per hit (O(n^2)); same result with one pass."""
    if cfg.DATA_CONFIG.DATASET == 'KittiDataset':
        lab_ids, lab_infos = list(strategy.labelled_set.sample_id_list), list(strategy.labelled_set.kitti_infos)
    else:
        lab_ids, lab_infos = list(strategy.labelled_set.frame_ids), list(strategy.labelled_set.infos)
    chosen = set(selected_frames)
    un_ids, un_infos = [], []
    for fid, info in strategy.pairs:
        if fid in chosen:
            lab_ids.append(fid)
            lab_infos.append(info)
        else:
            un_ids.append(fid)
            un_infos.append(info)
    return [tuple(lab_ids), tuple(lab_infos), tuple(un_ids), tuple(un_infos)]


def select_active_labels(model, labelled_loader, unlabelled_loader, rank, logger, method, leave_pbar=True, cur_epoch=None,
                         dist_train=False, active_label_dir=None, accumulated_iter=None):
    strategy = query_strategies.build_strategy(method=method, model=model, labelled_loader=labelled_loader,
                                               unlabelled_loader=unlabelled_loader, rank=rank,
                                               active_label_dir=active_label_dir, cfg=cfg)
    resume = os.path.join(active_label_dir, 'selected_frames_epoch_{}.pkl'.format(cur_epoch))
    if os.path.isfile(resume):
        print('found {} epoch saved selections...start resuming...'.format(cur_epoch))
        with open(resume, 'rb') as f:
            selected_frames = pkl.load(f)      # the reference calls Unpickler.load(f) with a stray argument (TypeError)
    else:
        selected_frames = strategy.query(leave_pbar, cur_epoch)
        strategy.save_active_labels(selected_frames=selected_frames, cur_epoch=cur_epoch)
        strategy.update_dashboard(cur_epoch=cur_epoch, accumulated_iter=accumulated_iter)
    active_training = move_selected_frames(strategy, selected_frames)
    if hasattr(strategy, 'close'):
        strategy.close()

    batch_size = unlabelled_loader.batch_size
    print('Batch_size of a single loader: %d' % (batch_size))
    workers = unlabelled_loader.num_workers
    del labelled_loader, unlabelled_loader
    labelled_set, unlabelled_set, labelled_loader, unlabelled_loader, sampler_labelled, sampler_unlabelled = \
        build_active_dataloader(cfg.DATA_CONFIG, cfg.CLASS_NAMES, batch_size, dist_train, workers=workers, logger=logger,
                                training=True, active_training=active_training)
    return labelled_loader, unlabelled_loader
