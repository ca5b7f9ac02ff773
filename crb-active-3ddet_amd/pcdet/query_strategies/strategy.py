"""Strategy base class (pcdet/query_strategies/strategy.py:5-83): bookkeeping of the per-frame GT statistics and the
pickle written per selection round. wandb is optional (the reference imports it unconditionally)."""
import os
import pickle


class _ListBatchSampler:
    """batch sampler over an explicit list of index lists, replaced before every pass of a persistent-worker loader"""

    def __init__(self):
        self.batches = []

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def _single_threaded_worker(worker_id):
    """loader workers are the parallelism: one BLAS / OpenMP thread each (numpy's BLAS otherwise starts a thread per core in
    every worker — 48 workers on a 256-thread host ran 2.3x slower per frame than one worker alone)"""
    import torch
    torch.set_num_threads(1)
    try:
        from threadpoolctl import threadpool_limits
        global _BLAS_LIMIT
        _BLAS_LIMIT = threadpool_limits(limits=1)               # kept referenced for the life of the worker
    except Exception:
        pass


def merge_collated(parts):
    """collated sub-batches (DatasetTemplate.collate_batch layout, consecutive frames) -> the batch collate_batch would have
    produced for all their frames at once"""
    import numpy as np
    if len(parts) == 1:
        return parts[0]
    out, base = {}, 0
    keys = parts[0].keys()
    sizes = [p['batch_size'] for p in parts]
    for key in keys:
        vals = [p[key] for p in parts]
        if key == 'batch_size':
            out[key] = int(sum(sizes))
        elif key in ('points', 'voxel_coords'):
            shifted, base = [], 0
            for v, b in zip(vals, sizes):
                v = v.copy() if base else v
                if base:
                    v[:, 0] += base
                shifted.append(v)
                base += b
            out[key] = np.concatenate(shifted, axis=0)
        elif key == 'point_frame_offsets':
            offs, base = [vals[0]], int(vals[0][-1])
            for v in vals[1:]:
                offs.append(v[1:] + base)
                base += int(v[-1])
            out[key] = np.concatenate(offs).astype(np.int32)
        elif key == 'gt_boxes':
            mx = max(v.shape[1] for v in vals)
            g = np.zeros((sum(sizes), mx, vals[0].shape[-1]), dtype=np.float32)
            r = 0
            for v in vals:
                g[r:r + v.shape[0], :v.shape[1]] = v
                r += v.shape[0]
            out[key] = g
        elif isinstance(vals[0], np.ndarray):
            out[key] = np.concatenate(vals, axis=0)
        else:
            out[key] = vals[0]
    return out


class Strategy:
    def __init__(self, model, labelled_loader, unlabelled_loader, rank, active_label_dir, cfg):
        self.cfg = cfg
        self.active_label_dir = active_label_dir
        self.rank = rank
        self.model = model
        self.labelled_loader = labelled_loader
        self.unlabelled_loader = unlabelled_loader
        self.labelled_set = labelled_loader.dataset
        self.unlabelled_set = unlabelled_loader.dataset
        self.bbox_records = {}
        self.point_measures = ['mean', 'median', 'variance']
        for met in self.point_measures:
            setattr(self, '{}_point_records'.format(met), {})
        ds = self.unlabelled_set
        if cfg.DATA_CONFIG.DATASET == 'KittiDataset':
            self.pairs = list(zip(ds.sample_id_list, ds.kitti_infos))
        else:
            self.pairs = list(zip(ds.frame_ids, ds.infos))

    POOL_SUB_BATCH = 2        # frames per loader work item

    def iter_pool_batches(self, frame_indices, batch_size):
        """host batches (collated like the loader's) of the given pool frames, in order. Frames are read and collated by as
        many DataLoader workers as the caller gave `unlabelled_loader` (the reference iterates that loader itself,
        crb_sampling.py:72-80), so reading / decoding the next frames overlaps the GPU work on the current ones; with
        num_workers == 0 the frames are read inline. A work item is a SUB-batch of POOL_SUB_BATCH frames, merged here: with
        whole batches as work items every worker starts on a batch of its own and the first one is complete only after a
        worker has read 16 frames alone (1.3 s of an idle GPU at the start of every pass on the synthetic pool), with
        single frames the per-item cost of the result queue dominates. The worker processes are started once per strategy
        and reused by every call (stage 1, stage 2, ...): forking them costs 2-3 s each time."""
        ds = self.unlabelled_set
        frame_indices = list(frame_indices)
        workers = int(getattr(self.unlabelled_loader, 'num_workers', 0) or 0)
        if not getattr(ds, 'device_voxelize', True):
            workers = 0                     # a host-voxelising dataset calls the HIP voxelizer: not from forked workers
        batches = [frame_indices[s:s + batch_size] for s in range(0, len(frame_indices), batch_size)]
        if workers > 0 and len(batches) > 1:
            sub = max(1, min(self.POOL_SUB_BATCH, batch_size))
            items, per_batch = [], []
            for chunk in batches:
                parts = [chunk[s:s + sub] for s in range(0, len(chunk), sub)]
                items += parts
                per_batch.append(len(parts))
            if getattr(self, '_pool_loader', None) is None:
                from torch.utils.data import DataLoader
                self._pool_batches = _ListBatchSampler()
                self._pool_loader = DataLoader(ds, batch_sampler=self._pool_batches, num_workers=workers,
                                               collate_fn=ds.collate_batch, pin_memory=False, prefetch_factor=4,
                                               persistent_workers=True, worker_init_fn=_single_threaded_worker)
            self._pool_batches.batches = items
            it = iter(self._pool_loader)
            for k in per_batch:
                yield merge_collated([next(it) for _ in range(k)])
            return
        for chunk in batches:
            yield ds.collate_batch([ds[i] for i in chunk])

    def close(self):
        """stop the pool loader's worker processes (they otherwise live as long as the strategy object)"""
        loader = getattr(self, '_pool_loader', None)
        if loader is not None:
            it = getattr(loader, '_iterator', None)
            if it is not None and hasattr(it, '_shutdown_workers'):
                it._shutdown_workers()
            self._pool_loader = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ distributed helpers / GT statistics
    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    @property
    def detector(self):
        return getattr(self.model, 'module', self.model)          # DDP-wrapped or bare

    @property
    def layout(self):
        """fixed-stride record layout of this detector (scoring.RecordLayout.for_model)"""
        if getattr(self, '_layout', None) is None:
            from . import scoring
            self._layout = scoring.RecordLayout.for_model(self.detector)
        return self._layout

    def record_gt_stats(self, gt_stats, frame_ids=None):
        """gt_stats (F, C, 5) rows {num_bbox, n_counted, mean, median, variance} in pool order (device or host) ->
        save_points() for every frame: what the reference's loops do frame by frame with pred_dicts[b] (crb_sampling.py:84,
        entropy_sampling.py:41, random_sampling.py:41), here once for the whole all-gathered pool so that EVERY rank can
        answer save_active_labels() for any selected frame."""
        from ..models.detectors.post_processing import class_names_of, gt_stats_to_dicts
        names = class_names_of(self.detector)
        host = gt_stats.detach().float().cpu().numpy() if hasattr(gt_stats, 'detach') else gt_stats
        ids = frame_ids if frame_ids is not None else [p[0] for p in self.pairs]
        assert len(ids) == host.shape[0], (len(ids), host.shape)
        for fid, row in zip(ids, host):
            nb, mean_p, med_p, var_p = gt_stats_to_dicts(row, names)
            self.save_points(fid, {'num_bbox': nb, 'mean_points': mean_p, 'median_points': med_p,
                                   'variance_points': var_p})

    def gt_stats_pool(self, frame_indices, batch_size):
        """GT point statistics of the given pool frames WITHOUT a detector pass (they depend on points and gt boxes only):
        -> (len, C, 5) device tensor"""
        import torch
        from ..models import load_data_to_gpu
        from ..models.detectors.post_processing import class_names_of, gt_point_stats_device
        nc = len(class_names_of(self.detector))
        rows = []
        for batch in self.iter_pool_batches(frame_indices, batch_size):
            load_data_to_gpu(batch)
            batch['point_frame_offsets'] = batch['point_frame_offsets'].int()
            rows.append(gt_point_stats_device(batch, nc)[0])
        if not rows:
            dev = next(self.detector.parameters()).device
            return torch.zeros((0, nc, 5), dtype=torch.float32, device=dev)
        return torch.cat(rows, 0)

    def save_points(self, frame_id, batch_dict):
        self.bbox_records[frame_id] = batch_dict['num_bbox']
        self.mean_point_records[frame_id] = batch_dict['mean_points']
        self.median_point_records[frame_id] = batch_dict['median_points']
        self.variance_point_records[frame_id] = batch_dict['variance_points']

    def update_dashboard(self, cur_epoch=None, accumulated_iter=None):
        try:
            import wandb
        except ImportError:
            return
        classes = list(self.selected_bbox[0].keys())
        total_bbox = 0
        for cls_idx in classes:
            num_cls_bbox = sum([i[cls_idx] for i in self.selected_bbox])
            wandb.log({'active_selection/num_bbox_{}'.format(cls_idx): num_cls_bbox}, step=accumulated_iter)
            total_bbox += num_cls_bbox
            for met in self.point_measures:
                rec = getattr(self, 'selected_{}_points'.format(met))
                val = sum([i[cls_idx] for i in rec]) / len(rec) if num_cls_bbox else 0
                wandb.log({'active_selection/{}_points_{}'.format(met, cls_idx): val}, step=accumulated_iter)
        wandb.log({'active_selection/total_bbox_selected': total_bbox}, step=accumulated_iter)

    def save_active_labels(self, selected_frames=None, grad_embeddings=None, cur_epoch=None):
        if selected_frames is not None:
            self.selected_bbox = [self.bbox_records[i] for i in selected_frames]
            for met in self.point_measures:
                setattr(self, 'selected_{}_points'.format(met),
                        [getattr(self, '{}_point_records'.format(met))[i] for i in selected_frames])
            path = os.path.join(self.active_label_dir,
                                'selected_frames_epoch_{}_rank_{}.pkl'.format(cur_epoch, self.rank))
            with open(path, 'wb') as f:
                pickle.dump({'frame_id': selected_frames, 'selected_mean_points': self.selected_mean_points,
                             'selected_bbox': self.selected_bbox,
                             'selected_median_points': self.selected_median_points,
                             'selected_variance_points': self.selected_variance_points}, f)
            print('successfully saved selected frames for epoch {} for rank {}'.format(cur_epoch, self.rank))
        if grad_embeddings is not None:
            with open(os.path.join(self.active_label_dir, 'grad_embeddings_epoch_{}.pkl'.format(cur_epoch)), 'wb') as f:
                pickle.dump(grad_embeddings, f)
            print('successfully saved grad embeddings for epoch {}'.format(cur_epoch))

    def query(self, leave_pbar=True, cur_epoch=None):
        pass
