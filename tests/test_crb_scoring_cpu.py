"""CPU: host logic of the CRB scoring path — record packing, rank-strided sharding + all-gather (2-process gloo), the
stage-3 prior and the stage-1 entropy against the oracle (reference library calls)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import crb_oracle


def test_pack_unpack_records_roundtrip():
    from pcdet.query_strategies import scoring
    B, P, R = 3, 128, 128
    g = torch.Generator().manual_seed(0)
    rec = {'entropy': torch.rand(B, generator=g), 'num': torch.tensor([5, 0, 128]),
           'pred_labels': torch.randint(1, 4, (B, P), generator=g), 'density': torch.rand((B, P), generator=g) * 100,
           'batch_rcnn_cls': torch.rand((B, R, 1), generator=g), 'batch_rcnn_reg': torch.randn((B, R, 7), generator=g),
           'gt_stats': torch.rand((B, 3, 5), generator=g)}
    rows = scoring.pack_records(rec)
    assert rows.shape == (B, scoring.REC_STRIDE) and scoring.REC_STRIDE == 1282 + 15
    u = scoring.unpack_records(rows)
    assert torch.equal(u['gt_stats'], rec['gt_stats'])
    assert torch.equal(u['num'], rec['num']) and torch.equal(u['labels'], rec['pred_labels'])
    assert torch.equal(u['density'], rec['density']) and torch.equal(u['rcnn_cls'], rec['batch_rcnn_cls'])
    assert torch.equal(u['rcnn_reg'], rec['batch_rcnn_reg']) and torch.equal(u['entropy'], rec['entropy'])


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'crb-active-3ddet_amd'))
    from pcdet.query_strategies import scoring
    idx, per = scoring.shard_indices(n, rank, world)
    local = torch.tensor([[float(i), float(i) * 2 + 1] for i in idx])          # row payload identifies the frame
    full = scoring.all_gather_rows(local, n, world)
    q.put((rank, full.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [10, 7])
def test_all_gather_rows_two_ranks_gloo(n):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    exp = np.array([[i, 2 * i + 1] for i in range(n)], np.float32)
    for rank, full in res:
        np.testing.assert_array_equal(full, exp)          # every rank holds the whole pool in dataset order


@pytest.mark.parametrize('n', [3000, 2999, 5])
def test_all_gather_rows_eight_ranks_gloo(n):
    """BASELINE configs[3] at its real rank count: 8 ranks, the 3,000-frame pool (375 frames per rank, no padding), a pool that does
    not divide (2,999: one wrap-around duplicate, dropped after the gather - pcdet/datasets/__init__.py:40), and a pool smaller
    than the world (5 frames on 8 ranks: three ranks score only wrap-around frames). Every rank ends with the whole pool in dataset
    order; the shard of rank r is r, r + 8, ... exactly as the reference's eval DistributedSampler deals the frames."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'crb-active-3ddet_amd'))
    from pcdet.query_strategies import scoring
    world = 8
    per = (n + world - 1) // world
    seen = []
    for r in range(world):
        idx, p = scoring.shard_indices(n, r, world)
        assert p == per and len(idx) == per
        assert idx == [(r + k * world) if (r + k * world) < n else (r + k * world) - n for k in range(per)]
        seen += idx
    assert sorted(set(seen)) == list(range(n))                       # every frame is scored by some rank
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    exp = np.array([[i, 2 * i + 1] for i in range(n)], np.float32)
    assert sorted(r for r, _ in res) == list(range(world))
    for rank, full in res:
        np.testing.assert_array_equal(full, exp)


def test_density_prior_matches_reference_formula():
    from pcdet.query_strategies import scoring
    rng = np.random.default_rng(0)
    dens = torch.from_numpy(np.concatenate([rng.gamma(2.0, 40.0, 900), rng.gamma(3.0, 150.0, 400),
                                            rng.gamma(2.0, 90.0, 300)]).astype(np.float32))
    lab = torch.from_numpy(np.concatenate([np.full(900, 1), np.full(400, 2), np.full(300, 3)]))
    perm = torch.randperm(len(lab), generator=torch.Generator().manual_seed(1))
    dens, lab = dens[perm], lab[perm]
    xa, pr = scoring.density_prior(dens, lab, 3)
    rx, rp = crb_oracle.build_prior(dens, lab, 3)
    np.testing.assert_array_equal(xa, np.stack(rx))
    np.testing.assert_allclose(pr, np.stack(rp), rtol=1e-15)


def test_label_entropy_matches_reference_formula():
    from pcdet.models.detectors.post_processing import label_entropy
    labs = torch.tensor([[1, 1, 2, 3, 0, 0], [2, 2, 2, 2, 2, 2], [0, 0, 0, 0, 0, 0], [1, 3, 3, 3, 0, 0]])
    valid = labs > 0
    got = label_entropy(labs, valid, 3)
    for b in range(4):
        exp = crb_oracle.label_entropy(labs[b][valid[b]], 3)
        assert abs(float(got[b]) - exp) < 1e-6


@pytest.mark.parametrize('n,d,k,seed', [(200, 512, 60, 0), (500, 4096, 300, 1), (64, 33, 40, 2), (300, 1000, 7, 3)])
def test_kmeans_plusplus_device_picks_equal_sklearn(n, d, k, seed):
    """scoring.kmeans_plusplus_device (ACTIVE_TRAIN.ACTIVE_CONFIG.CLUSTERING = 'kmeans++_device') restates sklearn's k-means++ seeding —
    the library call of the reference (crb_sampling.py:225-229) — with the same RandomState stream: identical picks, in
    order, on float32 embeddings of very different norms (here on CPU tensors; the GPU test repeats it on the device)"""
    from sklearn.cluster import kmeans_plusplus
    from pcdet.query_strategies import scoring
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) * rng.uniform(0.1, 3, size=(n, 1))).astype(np.float32)
    _, want = kmeans_plusplus(X, n_clusters=k, random_state=0)
    got = scoring.kmeans_plusplus_device(torch.from_numpy(X), k, random_state=0).numpy()
    np.testing.assert_array_equal(got, want)
    assert len(set(got.tolist())) == k


def test_pool_batches_from_persistent_loader_workers():
    """Strategy.iter_pool_batches: frames collated by the unlabelled loader's worker processes (started once, re-used by the
    next pass with another index list) == frames read inline, in the requested order"""
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.query_strategies.strategy import Strategy
    ds = SyntheticDataset(num_frames=10, n_points=1500)
    cfg = pv_rcnn_cfg()
    inline = Strategy(None, build_synthetic_dataloader(ds, 2), build_synthetic_dataloader(ds, 4), 0, '/tmp', cfg)
    st = Strategy(None, build_synthetic_dataloader(ds, 2), build_synthetic_dataloader(ds, 4, workers=2), 0, '/tmp', cfg)
    try:
        for idx, bs in (([9, 1, 3, 5, 7], 2), ([0, 2, 4, 6], 3), ([8], 4)):
            a = list(inline.iter_pool_batches(idx, bs))
            b = list(st.iter_pool_batches(idx, bs))
            assert len(a) == len(b) == (len(idx) + bs - 1) // bs
            for x, y in zip(a, b):
                assert list(x['frame_id']) == list(y['frame_id'])
                np.testing.assert_array_equal(np.asarray(x['points']), np.asarray(y['points']))
                np.testing.assert_array_equal(np.asarray(x['gt_boxes']), np.asarray(y['gt_boxes']))
        assert st._pool_loader is not None                      # workers kept between the passes
    finally:
        st.close()
    assert st._pool_loader is None
