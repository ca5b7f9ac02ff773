"""N > 1 paths with two processes on ONE device (gloo rendezvous on 127.0.0.1; RCCL needs one GPU per rank, which the test
box does not have): CRBSampling.query end to end with both all-gathers, and a 2-rank DDP SECOND step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, 'crb-active-3ddet_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    return torch.device('cuda', 0)


def _crb_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    dev = _setup(rank, world, port)
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy
    cfg = pv_rcnn_cfg()
    cfg.ACTIVE_TRAIN.SELECT_NUMS = 2
    cfg.ACTIVE_TRAIN.ACTIVE_CONFIG.K1 = 4            # 8 frames get gradient embeddings (4 per rank at world 2)
    cfg.ACTIVE_TRAIN.ACTIVE_CONFIG.K2 = 2
    cfg.ACTIVE_TRAIN.ACTIVE_CONFIG.FRAME_SEED = 1234
    torch.manual_seed(0)
    pool = SyntheticDataset(num_frames=11, first_frame=2100)          # 11 frames / 2 ranks: one wrap-around pad
    model = build_network(cfg.MODEL, 3, pool).to(dev)
    for m in model.modules():                        # deterministic scoring: MC dropout becomes the identity (the layers
        if isinstance(m, torch.nn.Dropout):          # stay in place: shared_fc_layer[4] must remain the second conv)
            m.p = 0.0
    with torch.no_grad():
        model.roi_head.cls_layers[-1].bias.fill_(1.0)
    strat = build_strategy('crb', model, build_synthetic_dataloader(SyntheticDataset(num_frames=2), 2),
                           build_synthetic_dataloader(pool, 3), rank, out_dir, cfg)
    picked = strat.query(cur_epoch=5)
    strat.save_active_labels(selected_frames=picked, cur_epoch=5)     # per-rank file, every rank can answer for any frame
    torch.save({'picked': picked, 'records': strat.last_records.cpu(), 'n_records': len(strat.bbox_records)},
               os.path.join(out_dir, 'w%d_r%d.pt' % (world, rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(fn, world, args):
    ctx = mp.get_context('spawn')
    port = 29500 + (os.getpid() * 7 + world * 131) % 3000
    procs = [ctx.Process(target=fn, args=(r, world, port) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0, 'worker failed (exit code %s)' % p.exitcode


def test_crb_query_two_ranks_equals_one_rank(tmp_path):
    out = str(tmp_path)
    _run(_crb_worker, 2, (out,))
    _run(_crb_worker, 1, (out,))
    r0, r1 = torch.load(os.path.join(out, 'w2_r0.pt')), torch.load(os.path.join(out, 'w2_r1.pt'))
    one = torch.load(os.path.join(out, 'w1_r0.pt'))
    assert r0['picked'] == r1['picked'] and len(r0['picked']) == 2          # identical picks on both ranks
    assert torch.equal(r0['records'], r1['records'])                         # the same gathered pool on both ranks
    assert r0['n_records'] == r1['n_records'] == 11
    # stage-1 records do not depend on how the pool was sharded (eval mode, per-frame quantities): integer fields exact,
    # float fields to f32 rounding of differently batched GEMMs
    from pcdet.query_strategies import scoring
    a, b = scoring.unpack_records(r0['records']), scoring.unpack_records(one['records'])
    assert torch.equal(a['num'], b['num']) and torch.equal(a['labels'], b['labels'])
    assert torch.equal(a['gt_stats'], b['gt_stats'])
    torch.testing.assert_close(a['entropy'], b['entropy'], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a['density'], b['density'], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a['rcnn_cls'], b['rcnn_cls'], rtol=1e-3, atol=1e-4)
    assert r0['picked'] == one['picked'], (r0['picked'], one['picked'])      # and so does the selection (FRAME_SEED set)
    for r in (0, 1):
        assert os.path.isfile(os.path.join(out, 'selected_frames_epoch_5_rank_%d.pkl' % r))


def _ddp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    dev = _setup(rank, world, port)
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(second_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0]) if world > 1 else model

    def batch(first):
        pts, off, gt = kitti_batch(first, 2, 6000)
        bidx = np.repeat(np.arange(2, dtype=np.float32), np.diff(off))
        return {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
                'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev),
                'batch_size': 2}
    grads = None
    for first in ([3000 + 2 * rank] if world > 1 else [3000, 3002]):          # world 1: both ranks' batches in turn
        model.zero_grad(set_to_none=True)
        ret, _, _ = net(batch(first))
        ret['loss'].mean().backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double()
        grads = g if grads is None else grads + g
    if world == 1:
        grads = grads / 2                                                    # DDP averages over ranks
    torch.save(grads.cpu(), os.path.join(out_dir, 'ddp_w%d_r%d.pt' % (world, rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_ddp_second_step_two_ranks_averages_gradients(tmp_path):
    out = str(tmp_path)
    _run(_ddp_worker, 2, (out,))
    _run(_ddp_worker, 1, (out,))
    g0, g1 = torch.load(os.path.join(out, 'ddp_w2_r0.pt')), torch.load(os.path.join(out, 'ddp_w2_r1.pt'))
    ref = torch.load(os.path.join(out, 'ddp_w1_r0.pt'))
    assert torch.equal(g0, g1)                                               # the all-reduce left identical gradients
    rel = float((g0 - ref).norm() / ref.norm())
    assert rel < 2e-3, rel              # = mean of the two single-rank gradients (MIOpen weight-gradient kernels use atomics)


def _syncbn_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    dev = _setup(rank, world, port)
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(second_cfg().MODEL, 3, SyntheticDataset(num_frames=2))
    model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model).to(dev)       # tools/train.py:168-169 (--sync_bn)
    n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
    model.train()
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
    pts, off, gt = kitti_batch(3100 + 2 * rank, 2, 6000)
    bidx = np.repeat(np.arange(2, dtype=np.float32), np.diff(off))
    b = {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
         'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': 2}
    ret, _, _ = net(b)
    loss = ret['loss'].mean()
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double()
    rm = model.backbone_2d.blocks[0][2].running_mean.detach().double().cpu()
    torch.save({'loss': float(loss), 'grads': g.cpu(), 'n_sync': n_sync, 'running_mean': rm}, os.path.join(out_dir, 'sbn_r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batchnorm_converted_mirror_trains_on_two_ranks(tmp_path):
    """`--sync_bn` of the reference's tools/train.py:168-169: SyncBatchNorm.convert_sync_batchnorm on the mirror replaces every
    BatchNorm1d / 2d; the fused BatchNorm pattern matches of the sparse / BEV modules (isinstance BatchNorm1d / 2d) then do not
    apply and the layers run as torch's SyncBatchNorm (statistics all-gathered over the ranks). One DDP step on two ranks:
    finite loss, identical all-reduced gradients, identical running statistics (the synchronised batch statistics)."""
    out = str(tmp_path)
    _run(_syncbn_worker, 2, (out,))
    r0, r1 = torch.load(os.path.join(out, 'sbn_r0.pt')), torch.load(os.path.join(out, 'sbn_r1.pt'))
    assert r0['n_sync'] == r1['n_sync'] and r0['n_sync'] >= 12 + 12                 # 12 sparse + 12 BEV layers at least
    assert np.isfinite(r0['loss']) and np.isfinite(r1['loss'])
    assert torch.isfinite(r0['grads']).all() and float(r0['grads'].abs().sum()) > 0
    assert torch.equal(r0['grads'], r1['grads'])
    torch.testing.assert_close(r0['running_mean'], r1['running_mean'], rtol=0, atol=0)


def test_bench_two_rank_dry_run_prints_the_contract_line(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* from the environment), except that both ranks share the one device of this box and the process
    group is gloo (CRB_DIST_BACKEND; RCCL needs one GPU per rank): DDP training steps, the rank-strided scoring shard, the
    records all-gather and the stage-2 embedding all-gather all execute, rank 0 prints ONE JSON line with the whole-job
    numbers and the per-rank diagnostics the first real SCALE run will need."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CRB_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    pool = 64
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--scoring-pool', str(pool), '--scoring-repeats', '1', '--pvrcnn-steps', '0', '--bf16x3-steps', '0',
           '--no-cpu-baseline', '--n1-value', '100.0']
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['higher_is_better']
    assert d['config']['global_batch'] == 32 and d['config']['parallelism'] == 'dp2'
    assert abs(d['value'] - 32 * 2 / (d['ms_per_step'] * 2 / 1e3)) < 1e-2 * d['value']
    assert len(d['ms_per_step_device_per_rank']) == 2 and all(v > 0 for v in d['ms_per_step_device_per_rank'])
    assert len(d['ms_per_step_per_rank']) == 2 and all(0 < v <= 1.05 * d['ms_per_step'] for v in d['ms_per_step_per_rank'])
    # the headline numbers sit at the head of the line and again at its very end (a log tail keeps the end)
    assert list(d.keys())[-1] == 'summary' and d['summary'] == d['headline']
    assert d['summary']['second_frames_per_s'] == d['value'] and abs(d['summary']['frames_per_s_per_gpu'] * 2 - d['value']) < 1e-2
    assert abs(d['summary']['scaling_efficiency'] - d['value'] / (2 * 100.0)) < 1e-3       # (--n1-value 100)
    assert d['summary']['crb_scoring_frames_per_s'] == d['crb_scoring']['value']
    sc = d['crb_scoring']
    assert sc['config']['frames_per_gpu'] == pool // 2 and sc['config']['pool_frames'] == pool and sc['config']['n_gpus'] == 2
    assert len(sc['per_rank_seconds']['loader_pass']) == 2
    coll = sc['collectives']
    strides = {c['bytes_per_rank'] // c['rows_per_rank'] for c in coll}
    assert len(coll) >= 3 and all(c['backend'] == 'gloo' for c in coll)
    assert sc['record_bytes_per_frame'] in strides                      # the stage-1 records all-gather
    assert 4 * 65536 in strides                                         # the stage-2 (256 x 256) embedding all-gather
    assert d['cpu_baseline'] is None and d['vs_baseline'] is None
    # first-real-run hygiene: every rank searched MIOpen solvers in its own user db, warm-up time is reported per rank
    assert len(d['warmup_seconds_per_rank']) == 2 and all(v > 0 for v in d['warmup_seconds_per_rank'])
    assert d['miopen_user_db'] == 'per rank'


# ---- RCCL on the one GPU of this box: a world-size-1 `nccl` process group loads librccl and executes the
#      collectives of the N > 1 code before the driver's 8-GPU node ever sees them

def _nccl1_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, 'crb-active-3ddet_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['CRB_FORCE_DIST'] = '1'                  # (read at import: the collectives run at world size 1 too)
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == 'nccl'
    from pcdet.query_strategies import scoring
    assert scoring.FORCE_COLLECTIVE
    # (a) the records all-gather: same rows back, through all_gather_into_tensor on RCCL
    scoring.COLLECTIVE_LOG = []
    local = torch.randn(7, 1287, device=dev)
    got = scoring.all_gather_rows(local, 5, 1)
    assert torch.equal(got, local[:5])
    emb = torch.randn(3, 65536, device=dev)
    assert torch.equal(scoring.all_gather_rows(emb, 3, 1), emb)
    log = scoring.COLLECTIVE_LOG
    assert len(log) == 2 and all(c['backend'] == 'nccl' for c in log) and log[1]['bytes_per_rank'] == 3 * 65536 * 4
    # all-reduce / barrier / all_gather list (what bench.py's timing brackets use)
    t = torch.tensor([3.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t) == 3.5
    # (b) one DDP step of SECOND over RCCL = the same step without the wrapper (one rank: the average of one gradient)
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(second_cfg().MODEL, 3, SyntheticDataset(num_frames=2)).to(dev)
    model.train()
    pts, off, gt = kitti_batch(3000, 2, 6000)
    bidx = np.repeat(np.arange(2, dtype=np.float32), np.diff(off))

    def batch():
        return {'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(dev),
                'point_frame_offsets': torch.from_numpy(off).to(dev), 'gt_boxes': torch.from_numpy(gt).to(dev), 'batch_size': 2}
    grads = []
    for net in (model, torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], gradient_as_bucket_view=True, bucket_cap_mb=4)):
        model.zero_grad(set_to_none=True)
        ret, _, _ = net(batch())
        ret['loss'].mean().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).double().cpu())
    torch.save({'plain': grads[0], 'ddp': grads[1]}, os.path.join(out_dir, 'nccl1.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_size_one_group_runs_the_collectives_and_a_ddp_step(tmp_path):
    out = str(tmp_path)
    _run(_nccl1_worker, 1, (out,))
    g = torch.load(os.path.join(out, 'nccl1.pt'))
    rel = float((g['plain'] - g['ddp']).norm() / g['plain'].norm())
    assert torch.isfinite(g['ddp']).all() and rel < 2e-3, rel           # (MIOpen's weight-gradient kernels use atomics: not bit-equal)


def test_bench_one_rank_over_rccl_prints_the_contract_line():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1` with CRB_FORCE_DIST=1: the process group is `nccl` (RCCL), the model is
    DDP-wrapped, the barriers / max-over-ranks reductions and the two scoring all-gathers execute on the GPU - the code the
    driver's N = 2, 4, 8 runs will execute, minus the second device."""
    import json
    import subprocess
    env = dict(os.environ, CRB_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('CRB_DIST_BACKEND', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29547', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--scoring-pool', '32', '--scoring-repeats', '1', '--pvrcnn-steps', '0', '--bf16x3-steps', '0', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0
    coll = d['crb_scoring']['collectives']
    assert len(coll) >= 2 and all(c['backend'] == 'nccl' for c in coll)
    assert 4 * 65536 in {c['bytes_per_rank'] // c['rows_per_rank'] for c in coll}
