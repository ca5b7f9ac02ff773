#!/usr/bin/env python
"""bench.py — SECOND fwd+bwd (+optimizer) on synthetic KITTI 20k-point clouds, one process per GPU.

Contract (see task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched by
torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the env). Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one batch of 16 frames per GPU (BASELINE.json configs[1]):
HIP voxelize(+mean) -> 12 sparse convs (HIP rulebooks + MFMA gather-GEMM) + BN/ReLU -> HIP dense scatter ->
BEV backbone (MIOpen) -> anchor head -> batched target assignment -> losses -> backward (HIP dgrad/wgrad) ->
grad-clip -> Adam. Inputs are resident in HBM before the timed region. N>1: DDP over RCCL, weak scaling.

Extra objects on the JSON line: `roofline` (dominant hand-written kernel: subm gather-GEMM, HIP events on the launch
stream inside the timed region, algorithmic bytes per SURVEY §8d) and `cpu_baseline` (oracle CPU port on a bounded
sample; rank 0, N=1 only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'crb-active-3ddet_amd'))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: dense f32-input MFMA (exact f32), the dtype this path computes in


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='frames per GPU')
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--kind', default='kitti', choices=['kitti', 'waymo'])
    ap.add_argument('--pool', type=int, default=2, help='distinct resident batches cycled through')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--nchw', action='store_true', help='A/B: NCHW memory format for the dense BEV part')
    ap.add_argument('--cpu-frames', type=int, default=8)
    ap.add_argument('--scoring-frames', type=int, default=32,
                    help='frames per GPU for the CRB stage-1 scoring measurement (0 = skip)')
    return ap.parse_args()


def make_batches(args, rank, device):
    from pcdet.datasets.synthetic import kitti_batch
    batches = []
    for k in range(args.pool):
        first = 1000 * rank + k * args.batch
        pts, off, gt = kitti_batch(first, args.batch, args.points, waymo=(args.kind == 'waymo'))
        bidx = np.repeat(np.arange(args.batch, dtype=np.float32), np.diff(off))
        pts5 = np.concatenate([bidx[:, None], pts], axis=1)
        batches.append({
            'points': torch.from_numpy(pts5).to(device),
            'point_frame_offsets': torch.from_numpy(off).to(device),
            'gt_boxes': torch.from_numpy(gt).to(device),
            'batch_size': args.batch,
        })
    return batches


def pmc_traffic(cin, cout):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (bench.py cannot collect counters on itself:
    they come from `bash tools/pmc_sparse_conv.sh`, summarised in profiles/r01_pmc_sparse_conv_fwd.json); None if the
    summary is for another kernel instance"""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_sparse_conv_fwd.json')
    try:
        d = json.load(open(path))
    except OSError:
        return None
    return d['traffic_bytes_per_launch_bench_mix'] if '<%d,%d>' % (cin, cout) in d.get('kernel', '') else None


def roofline_from_profile(prof):
    """dominant subm gather-GEMM instance by total time; algorithmic bytes per SURVEY §8d:
    B_alg = 4 N_in C_in + 4 N_out C_out + 8 P + 4 K C_in C_out"""
    agg = {}
    for kind, cin, cout, K, n_in, n_out, tab, e0, e1 in prof:
        if kind not in ('subm_fwd', 'subm_dgrad'):
            continue
        ms = e0.elapsed_time(e1)
        key = ('subm_gather_gemm', cin, cout)
        a = agg.setdefault(key, {'ms': 0.0, 'n': 0, 'bytes': 0.0, 'flops': 0.0, 'pairs': {}})
        tab = tab if not isinstance(tab, tuple) else tab[0]
        pid = tab.data_ptr()
        if pid not in a['pairs']:
            a['pairs'][pid] = int((tab >= 0).sum().item())
        P = a['pairs'][pid]
        a['ms'] += ms
        a['n'] += 1
        a['bytes'] += 4.0 * n_in * cin + 4.0 * n_out * cout + 8.0 * P + 4.0 * K * cin * cout
        a['flops'] += 2.0 * P * cin * cout
    if not agg:
        return None, {}
    key = max(agg, key=lambda k: agg[k]['ms'])
    a = agg[key]
    gbs = a['bytes'] / (a['ms'] * 1e-3) / 1e9
    table = {'%s_%dx%d' % k: {'launches': v['n'], 'avg_us': 1e3 * v['ms'] / v['n'],
                              'GBps_alg': v['bytes'] / (v['ms'] * 1e-3) / 1e9,
                              'TFLOPs': v['flops'] / (v['ms'] * 1e-3) / 1e12} for k, v in agg.items()}
    tfs = a['flops'] / (a['ms'] * 1e-3) / 1e12
    # which roof binds: the algorithmic bytes at 8 TB/s or the algorithmic flops on the exact-f32 MFMA
    # (v_mfma_f32_16x16x4_f32, 157.3 TF dense, MI355X_MICROARCH.md). At C=64 the intensity is ~117 flop/B against a
    # machine balance of 19.7, so the f32 gather-GEMM is MFMA-bound; the HBM fraction is reported beside it.
    t_hbm = a['bytes'] / (HBM_PEAK_GBS * 1e9)
    t_mfma = a['flops'] / (MFMA_F32_PEAK_TF * 1e12)
    kern = 'sparse_conv_fwd2_kernel<%d,%d> (subm gather-GEMM fwd+dgrad)' % (key[1], key[2])
    common = {'traffic': pmc_traffic(key[1], key[2]), 'avg_launch_us': round(1e3 * a['ms'] / a['n'], 2), 'launches': a['n'],
              'alg_bytes_per_launch': round(a['bytes'] / a['n']), 'alg_flops_per_launch': round(a['flops'] / a['n']),
              'hbm_GBps_alg': round(gbs, 1), 'hbm_frac': round(gbs / HBM_PEAK_GBS, 4),
              'mfma_f32_TFLOPs': round(tfs, 2), 'mfma_f32_frac': round(tfs / MFMA_F32_PEAK_TF, 4),
              'roof_us': {'hbm': round(1e6 * t_hbm / a['n'], 2), 'mfma_f32': round(1e6 * t_mfma / a['n'], 2)}}
    if t_mfma > t_hbm:
        roof = {'bound': 'mfma', 'kernel': kern, 'achieved': round(tfs, 2), 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': round(tfs / MFMA_F32_PEAK_TF, 4)}
    else:
        roof = {'bound': 'hbm', 'kernel': kern, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(gbs / HBM_PEAK_GBS, 4)}
    roof.update(common)
    return roof, table


def cpu_baseline(args):
    from pcdet.datasets import SyntheticDataset
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network
    from oracle.second_cpu import second_step_cpu
    # bounded sample: a few frames on at most 32 host threads (256-thread runs of this small problem are slower:
    # the first bench run took 152 s for 2 frames on 256 threads vs 8 s/frame on 8)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    os.environ['OMP_NUM_THREADS'] = str(cores)
    try:
        import ctypes
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(cores)
    except OSError:
        pass
    torch.manual_seed(0)
    ds = SyntheticDataset(num_frames=4, kind=args.kind, n_points=args.points)
    m = build_network(second_cfg(args.kind).MODEL, 3, ds)
    m.train()
    pts, off, gt = kitti_batch(0, args.cpu_frames, args.points, waymo=(args.kind == 'waymo'))
    t0 = time.time()
    second_step_cpu(m, pts, off, gt, max_voxels=ds.max_num_voxels['train'])
    dt = time.time() - t0
    return {'value': round(args.cpu_frames / dt, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': '%d synthetic %s frames x %d pts, one SECOND fwd+bwd step (oracle C voxelizer + sparse conv '
                      'fwd/dgrad/wgrad with OpenMP, stock torch CPU for BEV/head/loss), %.1f s' %
                      (args.cpu_frames, args.kind, args.points, dt)}


def crb_scoring_bench(args, rank, world, device):
    """CRB stage-1 acquisition scoring throughput: PV-RCNN eval forward with 5 MC-dropout head passes, batched
    post-processing records (final NMS, box point densities, label entropy) for a shard of `scoring_frames` frames per
    rank, then the RCCL all-gather of the fixed-stride records. Inputs resident in HBM; random-init weights."""
    from pcdet.datasets.synthetic import kitti_batch
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.models.detectors.post_processing import crb_frame_records
    from pcdet.query_strategies import scoring
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg()
    model = build_network(cfg.MODEL, 3, SyntheticDataset(num_frames=2)).to(device)
    model.eval()
    for m in model.modules():
        if m.__class__.__name__.startswith('Dropout'):
            m.train()
    bs = 16
    nb = max(1, args.scoring_frames // bs)
    batches = []
    for k in range(nb):
        pts, off, gt = kitti_batch(5000 + 1000 * rank + k * bs, bs, args.points)
        bidx = np.repeat(np.arange(bs, dtype=np.float32), np.diff(off))
        batches.append({'points': torch.from_numpy(np.concatenate([bidx[:, None], pts], 1)).to(device),
                        'point_frame_offsets': torch.from_numpy(off).to(device),
                        'gt_boxes': torch.from_numpy(gt).to(device), 'batch_size': bs,
                        'point_frame_counts_host': np.diff(off).tolist()})

    def run():
        rows = []
        with torch.no_grad():
            for b in batches:
                b = dict(b)
                model.pfe.prefetch_keypoints(b)          # as PVRCNN.forward does: FPS on a side stream
                for mod in model.module_list:
                    b = mod(b)
                rows.append(scoring.pack_records(crb_frame_records(model, b)))
        local = torch.cat(rows, 0)
        return scoring.all_gather_rows(local, local.shape[0] * world, world)

    run()                                   # warm-up (MIOpen solver search, allocator)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    rec = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = nb * bs * world
    return {'metric': 'frames/s CRB stage-1 acquisition scoring (PV-RCNN eval, 5 MC-dropout passes, records + all-gather)',
            'value': round(frames / dt, 3), 'unit': 'frames/s', 'frames': frames, 'seconds': round(dt, 3),
            'record_bytes_per_frame': 4 * scoring.REC_STRIDE, 'boxes_kept_total': int(rec[:, 1].sum().item())}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % ndev          # one process per GPU; the modulo only matters for the single-GPU gloo dry run
    torch.cuda.set_device(dev_idx)
    device = torch.device('cuda', dev_idx)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('CRB_DIST_BACKEND', 'nccl')          # 'nccl' = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    if os.environ.get('CRB_MIOPEN_FIND', '0') == '1':
        torch.backends.cudnn.benchmark = True       # MIOpen find mode: benchmark the applicable solvers per conv shape
    from crbhip import sparse as sp
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import second_cfg
    from pcdet.models import build_network

    if args.nchw:
        from pcdet.models.backbones_2d.map_to_bev import height_compression
        height_compression.CHANNELS_LAST = False
    torch.manual_seed(0)
    ds = SyntheticDataset(num_frames=args.batch, kind=args.kind, n_points=args.points)
    model = build_network(second_cfg(args.kind).MODEL, 3, ds).to(device)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=0.01, betas=(0.9, 0.99), fused=True)   # one multi-tensor kernel (0.23 vs 0.42 ms for the 84 tensors)
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_idx], gradient_as_bucket_view=True)
    batches = make_batches(args, rank, device)

    def step(i):
        b = dict(batches[i % len(batches)])
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = net(b)
        loss = ret['loss'].mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    prof = [] if rank == 0 else None
    sp.PROFILE = prof
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sp.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = args.batch * world * args.steps
    out = {
        'metric': 'frames/s SECOND fwd+bwd, KITTI 20k-pt clouds' if args.kind == 'kitti' else
                  'frames/s SECOND fwd+bwd, Waymo-shaped 160k-pt clouds',
        'value': round(frames / dt, 3), 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'SECOND (VoxelBackBone8x + BaseBEVBackbone + AnchorHeadSingle) fwd+bwd+AdamW on '
                               'synthetic %s clouds, %d pts/frame, bs=%d per GPU, HIP voxelize + subm/strided '
                               'gather-GEMM (BASELINE configs[1])' % (args.kind, args.points, args.batch),
                   'global_batch': args.batch * world, 'points_per_frame': args.points,
                   'parallelism': 'dp%d' % world, 'optimizer': 'AdamW in the timed region',
                   'final_loss': round(float(loss.item()), 4)},
    }
    del opt, net, model, batches
    torch.cuda.empty_cache()
    score = crb_scoring_bench(args, rank, world, device) if args.scoring_frames > 0 else None
    if rank == 0:
        out['crb_scoring'] = score
        roof, table = roofline_from_profile(prof)
        out['roofline'] = roof
        out['kernel_table'] = table
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
