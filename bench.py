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
import gc
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
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
MFMA_BF16_PEAK_TF = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md)


# gradient buckets of the DDP wrappers: SECOND has 21 MB of gradients, i.e. ONE default (25 MB) bucket whose all-reduce would
# start after the last backward kernel; 4 MB buckets go out while the BEV backbone's backward is still running (xGMI ring:
# ~30-50 us of latency per call, hidden) and leave only the sparse backbone's ~3 MB for the end of the step
DDP_BUCKET_MB = 4
# CRB_FORCE_DIST=1: run the N > 1 code (process group, DDP wrapper, barriers, score / embedding all-gathers) at world size 1 too:
# `torchrun --nproc-per-node 1 bench.py --gpus 1` then loads RCCL and executes every collective on the one GPU of a test box
FORCE_DIST = os.environ.get('CRB_FORCE_DIST', '0') == '1'


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
    ap.add_argument('--no-leg-tables', action='store_true', help='skip the per-launch kernel tables of the PV-RCNN and scoring legs')
    ap.add_argument('--n1-value', type=float, default=0.0, help='frames/s of the 1-GPU run of the same code: fills summary.scaling_efficiency '
                                                                'of an N > 1 line (value / (N x this))')
    ap.add_argument('--no-sparse-prefetch', action='store_true',
                    help='A/B: do not software-pipeline the next batch\'s voxel generator + table-plan marks behind this batch\'s forward '
                         'pass (Detector3DTemplate.prefetch_sparse); every step then reads the counts back in its sparse phase')
    ap.add_argument('--nchw', action='store_true', help='A/B: NCHW memory format for the dense BEV part')
    ap.add_argument('--bf16x3-steps', type=int, default=0, help='extra steps under the opt-in split-bf16 gather-GEMM (second roofline); 0 = skip')
    ap.add_argument('--miopen-steps', type=int, default=6, help='A/B: extra steps (and one resident scoring pass) with MIOpen\'s implicit GEMM '
                    'for the stride-1 3x3 BEV convolutions instead of the hand-written Winograd kernel that `value` runs (CRB_WINOGRAD=0; '
                    'reported beside `value`); 0 = skip')
    ap.add_argument('--cpu-frames', type=int, default=16, help='BASELINE configs[0]: 16 frames, one CPU fwd+bwd step')
    ap.add_argument('--scoring-pool', type=int, default=3000,
                    help='unlabeled pool size of the CRB stage-1 scoring measurement (BASELINE configs[3]: 3,000 frames, '
                         'rank-strided shard per GPU; 0 = skip)')
    ap.add_argument('--scoring-repeats', type=int, default=3)
    ap.add_argument('--stage2-batch', type=int, default=0, help='frames per train-mode stage-2 pass (0 = ACTIVE_CONFIG.STAGE2_BATCH)')
    ap.add_argument('--scoring-batch-ref', type=int, default=16, help='second scoring measurement at the reference\'s evaluation batch size '
                    '(reported as crb_scoring.at_reference_batch; 0 = skip)')
    ap.add_argument('--scoring-batch', type=int, default=64,
                    help='frames per scoring batch: an eval-mode pass scores every frame on its own, the batch size is the '
                         "caller's loader setting; ~4 ms of every pass are per-batch costs (table plan, small launches, idle): "
                         '16 -> 545, 32 -> 596, 64 -> 624, 128 -> 618 frames/s')
    ap.add_argument('--pvrcnn-steps', type=int, default=12, help='PV-RCNN fwd+bwd+AdamW steps (configs[2]; 0 = skip)')
    return ap.parse_args()


def make_batches(args, rank, device, first=0):
    from pcdet.datasets.synthetic import kitti_batch
    batches = []
    base = first
    for k in range(args.pool):
        first = base + 1000 * rank + k * args.batch
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
    they come from `bash tools/pmc_sparse_conv.sh`, summarised in profiles/r0N_pmc_sparse_conv_fwd.json, newest round first); None if the
    summary is for another kernel instance"""
    d = None
    for name in ('r03_pmc_sparse_conv_fwd.json', 'r02_pmc_sparse_conv_fwd.json', 'r01_pmc_sparse_conv_fwd.json'):
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            break
        except OSError:
            continue
    if d is None:
        return None
    return d['traffic_bytes_per_launch_bench_mix'] if '<%d,%d>' % (cin, cout) in d.get('kernel', '') else None


def event_pair_overhead_ms(n=200):
    """what an event pair reads with NOTHING between the two records (HIP event timestamps are taken by the command
    processor around the launch, so a pair around one short kernel over-reads by this much): median of n empty pairs"""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def other_kernel_rows(prof, overhead_ms=0.0):
    """rows of the kernel table for the hand-written kernels besides the subm gather-GEMM forward / input gradient, from the same
    HIP event pairs: Winograd convolutions (forward + input gradient: one kernel), Winograd weight gradient, sparse weight
    gradient, BatchNorm(+ReLU) forward / backward. Algorithmic work:
      Winograd conv   flops = 2 * 16 * tiles * Cin * Cout (the MFMA work of F(2x2,3x3): 16 GEMMs over the 2x2-output tiles;
                      the direct convolution it replaces is 2.25x that: `direct_equivalent_TFLOPs`), bytes = input + output map
                      + weight image;  weight gradient: the same flops, bytes = input + output-gradient map
      sparse wgrad    SURVEY 8d: B = 4 N_in C_in + 4 N_out C_out + 8 P + 4 K C_in C_out, F = 2 P C_in C_out
      BatchNorm       forward 3, backward 5 passes of 4 n C bytes"""
    agg = {}

    def add(key, ms, nbytes, flops, extra=None):
        a = agg.setdefault(key, {'ms': 0.0, 'n': 0, 'bytes': 0.0, 'flops': 0.0, 'extra': extra or {}})
        a['ms'] += ms
        a['n'] += 1
        a['bytes'] += nbytes
        a['flops'] += flops
    pairs_cache = {}
    for rec in prof:
        kind = rec[0]
        e0, e1 = rec[-2], rec[-1]
        ms = max(e0.elapsed_time(e1) - overhead_ms, 1e-4)
        if kind in ('wino_conv', 'wino_wgrad'):
            _, cin, cout, N, H, W = rec[:6]
            tiles = N * ((H + 1) // 2) * ((W + 1) // 2)
            fl = 2.0 * 16 * tiles * cin * cout
            px = float(N * H * W)
            by = 4.0 * px * (cin + cout) + (4.0 * 16 * cin * cout if kind == 'wino_conv' else 0.0)
            add(('winograd_conv' if kind == 'wino_conv' else 'winograd_wgrad', cin, cout, H, W), ms, by, fl,
                {'direct_flops': 2.0 * 9 * px * cin * cout})
        elif kind == 'wgrad':
            _, cin, cout, K, n_in, n_out, pstart = rec[:7]
            pid = pstart.data_ptr()
            if pid not in pairs_cache:
                pairs_cache[pid] = int(pstart[-1].item())
            P = pairs_cache[pid]
            add(('sparse_wgrad', cin, cout), ms, 4.0 * n_in * cin + 4.0 * n_out * cout + 8.0 * P + 4.0 * K * cin * cout, 2.0 * P * cin * cout)
        elif kind in ('bn_fwd', 'bn_bwd', 'bn_stats', 'bn_apply'):
            _, n, C = rec[:3]
            # bn_stats: statistics pass only (the apply lives in the next convolution's input transform, opt-in); bn_apply: apply
            # pass only (the statistics came from the producing convolution's epilogue)
            passes = {'bn_fwd': 3.0, 'bn_bwd': 5.0, 'bn_stats': 1.0, 'bn_apply': 2.0}[kind]
            add(('batchnorm_relu_' + kind[3:], C, 'rows>=1M' if n >= (1 << 20) else 'rows<1M'), ms, passes * 4.0 * n * C, 0.0)
    rows = {}
    for k, v in agg.items():
        name = '_'.join(str(x) for x in k)
        r = {'launches': v['n'], 'avg_us': round(1e3 * v['ms'] / v['n'], 2), 'ms_total': round(v['ms'], 3),
             'GBps_alg': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1), 'hbm_frac': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if v['flops'] > 0:
            tf = v['flops'] / (v['ms'] * 1e-3) / 1e12
            r.update({'TFLOPs': round(tf, 2), 'mfma_f32_frac': round(tf / MFMA_F32_PEAK_TF, 4)})
        if 'direct_flops' in v['extra']:
            r['direct_equivalent_TFLOPs'] = round(v['extra']['direct_flops'] * v['n'] / (v['ms'] * 1e-3) / 1e12, 1)
        r['_agg'] = v
        rows[name] = r
    return rows


def dominant_roofline(prof, overhead_ms, gather_roof, gather_table):
    """`roofline` = the hand-written kernel with the most time inside the timed steps (bench contract: the dominant kernel). Since
    round 4 that is the Winograd convolution kernel of the BEV backbone (40 % of a SECOND step), MFMA-bound; the subm gather-GEMM
    (north_star's named kernel) keeps its object as `roofline_gather_gemm`."""
    rows = other_kernel_rows(prof, overhead_ms)
    table = dict(gather_table)
    best, best_ms = None, -1.0
    for name, r in rows.items():
        v = r.pop('_agg')
        table[name] = r
        if name.startswith('winograd') and v['ms'] > best_ms:
            best, best_ms = (name, r, v), v['ms']
    g_ms = 0.0
    if gather_roof is not None:
        g_ms = gather_roof['avg_launch_us'] * gather_roof['launches'] * 1e-3
    if best is None or g_ms >= best_ms:
        return gather_roof, None, table
    name, r, v = best
    traffic, traffic_src = None, 'no committed PMC summary for this kernel instance'
    from crbhip import winograd as _w
    x6 = _w.KERNEL == 'x6' and name.startswith('winograd_conv')                # (the forward / input-gradient kernel of this run)
    for pmc in (('r06_pmc_winograd4.json',) if x6 else ('r05_pmc_winograd2.json', 'r04_pmc_winograd2.json')):   # newest committed PMC summary of this kernel
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', pmc)))
            hit = next((e for key, e in pm.get('shapes', {}).items() if name.endswith('conv_' + key)), None)   # (forward / input-gradient instances)
            if hit is not None:
                traffic, traffic_src = hit['traffic_bytes_per_launch'], pm['source']
                break
            if pm.get('shape') and name.endswith('conv_' + '_'.join(str(x) for x in pm['shape'])):
                traffic, traffic_src = pm['traffic_bytes_per_launch'], pm['source']
                break
        except (OSError, ValueError, KeyError):
            continue
    form_c = x6 and getattr(_w, 'FORM_C', False) and int(name.split('_')[3]) % 128 == 0       # winograd_conv_<cin>_<cout>_<H>_<W>
    kname = (('winograd4c_kernel' if form_c else 'winograd4_kernel') +
             ': %s (F(2x2,3x3), f32 in / f32 out, every product as six bf16 MFMA passes over an exact three-way split of '
             'the f32 operands, f32 accumulate; BEV backbone 3x3 convolutions)' if x6 else
             'winograd2_kernel / winograd2_wgrad_kernel: %s (F(2x2,3x3) f32 MFMA, BEV backbone 3x3 convolutions)') % name
    roof = {'bound': 'mfma', 'kernel': kname,
            'achieved': r['TFLOPs'], 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s', 'frac': r['mfma_f32_frac'],
            'traffic': traffic, 'traffic_source': traffic_src,
            'avg_launch_us': r['avg_us'], 'launches': r['launches'], 'event_pair_overhead_us': round(1e3 * overhead_ms, 2),
            'alg_flops_per_launch': round(v['flops'] / v['n']), 'alg_bytes_per_launch': round(v['bytes'] / v['n']),
            'hbm_GBps_alg': r['GBps_alg'], 'hbm_frac': r['hbm_frac'], 'direct_equivalent_TFLOPs': r.get('direct_equivalent_TFLOPs'),
            'roof_us': {'hbm': round(1e6 * v['bytes'] / v['n'] / (HBM_PEAK_GBS * 1e9), 2),
                        'mfma_f32': round(1e6 * v['flops'] / v['n'] / (MFMA_F32_PEAK_TF * 1e12), 2)},
            'note': 'flops = the MFMA work of the Winograd algorithm (16 GEMMs over 2x2-output tiles); the direct convolution it replaces '
                    'has 2.25x the flops (direct_equivalent_TFLOPs may exceed the MFMA peak: fewer multiplications, not a faster pipe)'}
    if x6:
        roof['bf16_mfma_TFLOPs_issued'] = round(6.0 * r['TFLOPs'], 1)
        roof['bf16_mfma_frac'] = round(6.0 * r['TFLOPs'] / MFMA_BF16_PEAK_TF, 4)
        roof['note'] += ('; achieved / peak / frac price the ALGORITHMIC f32 flops of the 16 GEMMs against the f32-input MFMA peak (the dtype of '
                         'the path: f32 operands, f32 results, errors against f64 at the f32-MFMA kernel\'s level); the matrix pipe issues six '
                         'bf16 passes per product: bf16_mfma_TFLOPs_issued against the 2,500 TFLOP/s dense bf16 peak = bf16_mfma_frac. The '
                         'kernel is bound by the latency of its in-order loads and the input transform, not by the matrix pipe (DESIGN section 6)')
    return roof, gather_roof, table


def roofline_from_profile(prof, overhead_ms=0.0, arithmetic='f32'):
    """dominant subm gather-GEMM instance by total time; algorithmic bytes per SURVEY §8d:
    B_alg = 4 N_in C_in + 4 N_out C_out + 8 P + 4 K C_in C_out. Launch durations = HIP event pairs on the launch stream
    inside the timed region minus the empty-pair reading (event_pair_overhead_ms). arithmetic='bf16x3': the launches of the
    opt-in split-bf16 kernel (its event pair also covers the W split kernel), priced against the HBM roof and the bf16 MFMA
    roof of its three passes."""
    agg = {}
    kinds = ('subm_fwd', 'subm_dgrad') if arithmetic == 'f32' else ('subm_fwd_bf16x3', 'subm_dgrad_bf16x3')
    for rec in prof:
        if rec[0] not in kinds:
            continue
        kind, cin, cout, K, n_in, n_out, tab, e0, e1 = rec
        ms = max(e0.elapsed_time(e1) - overhead_ms, 1e-4)
        key = ('subm_gather_gemm', cin, cout)
        a = agg.setdefault(key, {'ms': 0.0, 'n': 0, 'bytes': 0.0, 'flops': 0.0, 'pairs': {}})
        tab = tab if not isinstance(tab, tuple) else tab[0]
        compact = hasattr(tab, 'cbase')
        pid = tab.cbase.data_ptr() if compact else tab.data_ptr()
        if pid not in a['pairs']:
            a['pairs'][pid] = tab.num_pairs() if compact else int((tab >= 0).sum().item())
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
    if arithmetic == 'bf16x3':
        t_mfma3 = 3.0 * a['flops'] / (MFMA_BF16_PEAK_TF * 1e12)
        tf3 = 3.0 * tfs                                  # bf16 MFMA flops actually issued: three passes
        hbm_bound = t_hbm >= t_mfma3                     # the two roofs are within 20 % of each other at C = 64
        return {'bound': 'hbm' if hbm_bound else 'mfma',
                'kernel': 'sparse_conv_fwd_bf16x3_kernel<%d,%d> + w_split_pack_kernel (subm gather-GEMM fwd+dgrad, OPT-IN '
                          'split-bf16 contract: |y - y_f32| <= 2^-16 sum|x||w|)' % (key[1], key[2]),
                'achieved': round(gbs, 1) if hbm_bound else round(tf3, 1),
                'peak': HBM_PEAK_GBS if hbm_bound else MFMA_BF16_PEAK_TF, 'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
                'frac': round(gbs / HBM_PEAK_GBS, 4) if hbm_bound else round(tf3 / MFMA_BF16_PEAK_TF, 4),
                'hbm_GBps_alg': round(gbs, 1), 'hbm_frac': round(gbs / HBM_PEAK_GBS, 4),
                'mfma_bf16_TFLOPs_3pass': round(tf3, 1), 'mfma_bf16_frac': round(tf3 / MFMA_BF16_PEAK_TF, 4),
                'traffic': None, 'avg_launch_us': round(1e3 * a['ms'] / a['n'], 2), 'launches': a['n'],
                'event_pair_overhead_us': round(1e3 * overhead_ms, 2),
                'alg_bytes_per_launch': round(a['bytes'] / a['n']), 'alg_flops_per_launch': round(a['flops'] / a['n']),
                'roof_us': {'hbm': round(1e6 * t_hbm / a['n'], 2), 'mfma_bf16_3pass': round(1e6 * t_mfma3 / a['n'], 2)},
                'note': 'second roofline of the same launches (VERDICT r01 item 6); the default path and `value` stay exact f32'}, table
    t_mfma = a['flops'] / (MFMA_F32_PEAK_TF * 1e12)
    kern = 'sparse_conv_fwd2_kernel<%d,%d> (subm gather-GEMM fwd+dgrad)' % (key[1], key[2])
    common = {'traffic': pmc_traffic(key[1], key[2]),
              'traffic_source': 'separate rocprofv3 --pmc passes of the same kernel on the same tables '
                                '(tools/pmc_sparse_conv.sh -> profiles/*pmc_sparse_conv_fwd.json), not collected by this run',
              'event_pair_overhead_us': round(1e3 * overhead_ms, 2),
              'avg_launch_us': round(1e3 * a['ms'] / a['n'], 2), 'launches': a['n'],
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
    from pcdet.utils.common_utils import effective_cpu_count
    cores = min(effective_cpu_count(), 32)               # cgroup quota / affinity aware (the GPU pod: 16 of 256 threads)
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
    crb = cpu_baseline_crb(cores) if args.kind == 'kitti' else None
    return {'value': round(args.cpu_frames / dt, 4), 'unit': 'frames/s', 'cores': cores, 'kind': 'port', 'crb': crb,
            'threads_note': 'cores = the threads actually used = min(CPUs this process may use (cgroup quota / affinity: %d here), 32); '
                            'the cap of 32 exists because the oracle\'s OpenMP loops and torch\'s CPU convolutions of this 16-frame '
                            'problem get slower beyond that (first bench of r01: 152 s for 2 frames on 256 threads, 8 s per frame '
                            'on 8)' % effective_cpu_count(),
            'sample': '%d synthetic %s frames x %d pts (BASELINE configs[0]), one SECOND fwd+bwd step (oracle C voxelizer + '
                      'sparse conv fwd/dgrad/wgrad with OpenMP, functional torch CPU for BEV/head/targets/loss), %.1f s' %
                      (args.cpu_frames, args.kind, args.points, dt)}


def cpu_baseline_crb(cores, frames=48, picks=6):
    """CRB half of the CPU baseline ("CRB stage-1 scoring ... plus stage-3 KDE greedy"), with the very
    library calls the reference makes (oracle/crb_oracle.py: torch Categorical, sklearn KernelDensity, scipy entropy):
    stage-1 RECORDS of `frames` synthetic frames (label entropy, per-box point density through the oracle's
    points-in-boxes, per-class GT point statistics; crb_sampling.py:72-121 — the detector forward that produces the
    boxes is NOT in this figure: its CPU cost is the SECOND leg's scale) and the first `picks` greedy picks of stage 3 over
    K2*N = 300 candidates (crb_sampling.py:276-331), extrapolated to N = 100 picks with the last pick's time (the cost
    per pick grows with the selected set, so this is a lower bound)."""
    import oracle
    from oracle import crb_oracle
    from pcdet.datasets.synthetic import kitti_frame
    rng = np.random.default_rng(0)
    t0 = time.time()
    dens, labs = [], []
    for f in range(frames):
        pts, gt = kitti_frame(5000 + f, 20000)
        gt = gt[gt[:, 7] > 0]
        # detections: the frame's objects, jittered (what a trained detector returns), labels kept
        det = gt[:, :7] + rng.normal(0, 0.1, (len(gt), 7)).astype(np.float32) * np.array([1, 1, .3, .3, .3, .3, .2], np.float32)
        lab = torch.from_numpy(gt[:, 7].astype(np.int64))
        crb_oracle.label_entropy(lab, 3)
        idx = oracle.points_in_boxes(pts[None, :, :3].astype(np.float32), det[None].astype(np.float32))[0]
        cnt = np.bincount(idx[idx >= 0], minlength=len(det)).astype(np.float32)
        dens.append(torch.from_numpy(cnt / (det[:, 3] * det[:, 4] * det[:, 5])))
        labs.append(lab)
        crb_oracle.gt_point_statistics(pts[:, :3], gt, 3)
    t_rec = time.time() - t0
    # stage 3 over 300 candidates (the frames above, cycled) — the reference's O(picks x candidates x classes) KDE loop
    cand_d = [dens[i % frames] for i in range(300)]
    cand_l = [labs[i % frames] for i in range(300)]
    x_axis, prior = crb_oracle.build_prior(torch.cat(cand_d), torch.cat(cand_l), 3)
    ts = []
    for n_pick in (picks - 1, picks):
        t0 = time.time()
        crb_oracle.density_greedy(cand_d, cand_l, x_axis, prior, 3, n_pick)
        ts.append(time.time() - t0)
    per_pick = max(ts[1] - ts[0], 1e-9)
    stage3_est = ts[1] + per_pick * (100 - picks)
    return {'stage1_records_frames_per_s': round(frames / t_rec, 2), 'stage1_records_s_per_3000_frames': round(3000 * t_rec / frames, 1),
            'stage3_s_first_%d_picks' % picks: round(ts[1], 2), 'stage3_s_per_pick_at_%d' % picks: round(per_pick, 3),
            'stage3_s_100_picks_extrapolated': round(stage3_est, 1), 'cores': cores, 'kind': 'port',
            'sample': '%d synthetic KITTI frames x 20000 pts: stage-1 records (entropy, box point density, GT point '
                      'statistics; detector forward excluded) %.1f s; stage 3: %d of 100 greedy picks over 300 candidates '
                      '%.1f s, extrapolated linearly with the last pick' % (frames, t_rec, picks, ts[1])}


def _pctl(xs):
    xs = np.asarray(xs, dtype=np.float64)
    return {'median': round(float(np.median(xs)), 3), 'p10': round(float(np.percentile(xs, 10)), 3),
            'p90': round(float(np.percentile(xs, 90)), 3)}


def _all_ranks(x, world, device):
    """the value of every rank, in rank order (diagnostics of the first multi-GPU runs: which rank is slow)"""
    if world == 1 and not FORCE_DIST:
        return [float(x)]
    t = torch.tensor([float(x)], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def _max_over_ranks(dt, world, device):
    if world > 1 or FORCE_DIST:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


class CallProfiler:
    """HIP-event pairs around EVERY launch that goes through the C-ABI while it is active (the PV-RCNN and scoring
    legs on the driver's roofline). `lib.crb_*` attributes are replaced by wrappers that record an event on torch's current stream
    (= the stream handed to the call) before and after; run OUTSIDE the timed regions. Rows: per entry point and shape key - launches
    per unit (step / pass), average microseconds, and for the kernels with a work formula the algorithmic flops over LIVE rows
    (queries whose ball is not empty: read from the call's own empty mask) against the f32-MFMA peak, or the algorithmic bytes
     against HBM."""

    @staticmethod
    def _live(ptr_value, count):
        if not ptr_value or count <= 0:
            return None

        class _V:
            __cuda_array_interface__ = {'data': (int(ptr_value), False), 'shape': (int(count),), 'typestr': '|u1', 'version': 2}
        return (torch.as_tensor(_V(), device='cuda') == 0).sum()

    WORK = {
        # name: args -> (shape key, flops over all rows, bytes, (empty-mask pointer, rows) or None, extra)
        'crb_sa_mlp2_train_stats': lambda a: ((a[3], a[4], a[2]), 2.0 * a[1] * a[2] * a[3] * a[4], 0.0, (a[11], a[1]), {}),
        'crb_sa_mlp2_train_max': lambda a: ((a[3], a[4], a[2]), 2.0 * a[1] * a[2] * a[3] * a[4], 0.0, (a[11], a[1]), {}),
        'crb_sa_mlp2_train_backward': lambda a: ((a[3], a[4], a[2]), 6.0 * a[1] * a[2] * a[3] * a[4], 0.0, (a[11], a[1]), {}),
        'crb_sa_mlp2_max_stack': lambda a: ((a[3], a[4], a[2]), 2.0 * a[1] * a[2] * a[3] * a[4], 0.0, (a[11], a[1]), {}),
        'crb_group_affine_rows_stats_stack': lambda a: ((a[2], a[3]), 0.0, 4.0 * a[1] * a[3] * (1 + a[2]), (a[10], a[1]), {}),
        'crb_group_affine_rows_grad_bn_recompute_stack': lambda a: ((a[2], a[3]), 0.0, 4.0 * a[1] * a[3] * (1 + 2 * a[2]), (a[10], a[1]), {}),
        'crb_group_affine_rows_grad_stack': lambda a: ((a[2], a[3]), 0.0, 4.0 * a[1] * a[3] * (1 + 2 * a[2]), (a[7], a[1]), {}),
        'crb_farthest_point_sample': lambda a: ((a[1], a[2]), 0.0, 0.0, None, {'rounds': float(a[0]) * 0 + float(a[2] - 1), 'frames': a[0]}),
        'crb_ball_query2_stack': lambda a: ((a[3], a[5]), 0.0, 0.0, None, {'queries': float(a[1])}),
        'crb_ball_query2_grouped_stack': lambda a: ((a[2], a[4], a[6]), 0.0, 0.0, None, {'queries': float(a[1])}),
        'crb_nms_batched': lambda a: ((a[3], a[6]), 0.0, float(a[2]) * (28.0 * a[3] + 8.0 * a[3] * ((a[3] + 63) // 64)), None, {}),
    }

    def __init__(self):
        self.records = []
        self._saved = {}

    def __enter__(self):
        from crbhip import _lib
        L = _lib.lib
        for name in _lib.parse_header():
            fn = getattr(L, name, None)
            if fn is None or not fn.argtypes or name.endswith(('_bytes', '_supported', '_floats', '_waves', '_slabs')):
                continue
            self._saved[name] = fn

            def wrap(*args, _fn=fn, _name=name):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*args)
                e1.record()
                work = self.WORK.get(_name)
                key, fl, by, live, extra = (None, 0.0, 0.0, None, {})
                if work is not None:
                    vals = [getattr(v, 'value', v) for v in args]
                    key, fl, by, live, extra = work(vals)
                    if live is not None:
                        live = (self._live(live[0], live[1]), live[1])
                self.records.append((_name, key, fl, by, live, extra, e0, e1))
                return rc
            setattr(L, name, wrap)
        return self

    def __exit__(self, *exc):
        from crbhip import _lib
        for name, fn in self._saved.items():
            setattr(_lib.lib, name, fn)
        return False

    def rows(self, units, overhead_ms=0.0):
        """-> ({row name: row}, roofline of the row with the most time among the kernels that have a work formula)"""
        torch.cuda.synchronize()
        agg = {}
        for name, key, fl, by, live, extra, e0, e1 in self.records:
            ms = max(e0.elapsed_time(e1) - overhead_ms, 1e-4)
            frac = 1.0
            if live is not None and live[0] is not None:
                frac = float(live[0].item()) / max(1, live[1])
            k = name[4:] + ('' if key is None else '_' + '_'.join(str(x) for x in key))
            a = agg.setdefault(k, {'ms': 0.0, 'n': 0, 'flops': 0.0, 'bytes': 0.0, 'live': 0.0, 'rounds': 0.0, 'queries': 0.0})
            a['ms'] += ms; a['n'] += 1; a['flops'] += fl * frac; a['bytes'] += by * frac; a['live'] += frac
            a['rounds'] += extra.get('rounds', 0.0); a['queries'] += extra.get('queries', 0.0)
        rows, best = {}, None
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
            r = {'launches_per_unit': round(a['n'] / units, 2), 'avg_us': round(1e3 * a['ms'] / a['n'], 2), 'ms_per_unit': round(a['ms'] / units, 3)}
            if a['flops'] > 0:
                tf = a['flops'] / (a['ms'] * 1e-3) / 1e12
                r.update({'TFLOPs_live_rows': round(tf, 2), 'mfma_f32_frac': round(tf / MFMA_F32_PEAK_TF, 4), 'live_fraction': round(a['live'] / a['n'], 3)})
            if a['bytes'] > 0:
                gb = a['bytes'] / (a['ms'] * 1e-3) / 1e9
                r.update({'GBps_alg': round(gb, 1), 'hbm_frac': round(gb / HBM_PEAK_GBS, 4), 'live_fraction': round(a['live'] / a['n'], 3)})
            if a['rounds'] > 0:
                r['us_per_round'] = round(1e3 * a['ms'] / a['rounds'], 3)
            if a['queries'] > 0:
                r['Mqueries_per_s'] = round(a['queries'] / (a['ms'] * 1e-3) / 1e6, 1)
            rows[k] = r
            if (a['flops'] > 0 or a['bytes'] > 0) and (best is None or a['ms'] > best[1]['ms']):
                best = (k, a, r)
        roof = None
        if best is not None:
            k, a, r = best
            if a['flops'] > 0:
                roof = {'bound': 'mfma', 'kernel': k, 'achieved': r['TFLOPs_live_rows'], 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s',
                        'frac': r['mfma_f32_frac'], 'traffic': None, 'avg_launch_us': r['avg_us'], 'launches_per_unit': r['launches_per_unit'],
                        'alg_flops_per_launch': round(a['flops'] / a['n']), 'live_fraction': r['live_fraction'],
                        'note': 'the leg\'s own kernel with the most time; flops over live rows (queries with a non-empty ball)'}
            else:
                roof = {'bound': 'hbm', 'kernel': k, 'achieved': r['GBps_alg'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': r['hbm_frac'],
                        'traffic': None, 'avg_launch_us': r['avg_us'], 'launches_per_unit': r['launches_per_unit'],
                        'alg_bytes_per_launch': round(a['bytes'] / a['n'])}
        return rows, roof


def crb_scoring_bench(args, rank, world, device):
    """CRB stage-1 acquisition scoring over the BASELINE configs[3] pool (3,000 synthetic KITTI frames): every rank scores
    its rank-strided shard with the PV-RCNN eval forward + 5 MC-dropout head passes and the batched post-processing records
    (final NMS, box point densities, label entropy, GT point statistics), ONE all-gather of the fixed-stride rows (RCCL at
    N > 1), then the per-frame GT-statistics bookkeeping of the whole pool on every rank (what the caller pickles after the
    query). This is CRBSampling.stage1() — the code path of query() — timed two ways:
      loader    frames generated / collated by the unlabelled loader's worker processes and uploaded inside the timed pass
      resident  the shard's batches already in HBM (the `value`; median of --scoring-repeats passes)"""
    from pcdet.datasets import SyntheticDataset, build_synthetic_dataloader
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    from pcdet.query_strategies import build_strategy, scoring
    torch.manual_seed(0)
    cfg = pv_rcnn_cfg()
    n = args.scoring_pool
    bs = args.scoring_batch
    pool = SyntheticDataset(num_frames=n, first_frame=5000, n_points=args.points, training=False)
    lab = SyntheticDataset(num_frames=2, n_points=args.points)
    model = build_network(cfg.MODEL, 3, pool).to(device)
    from pcdet.utils.common_utils import effective_cpu_count
    # loader workers: the CPUs this process may use (cgroup quota / affinity, not os.cpu_count()), shared by the ranks, two
    # left for the process that feeds the GPU
    workers = max(2, min(48, effective_cpu_count() // max(world, 1) - 2))
    host_threads = torch.get_num_threads()
    torch.set_num_threads(2)                              # the main process: collate / pinned copies only, the workers get the cores
    strat = build_strategy('crb', model, build_synthetic_dataloader(lab, 2),
                           build_synthetic_dataloader(pool, bs, workers=workers), rank, '/tmp', cfg)
    mine, per = scoring.shard_indices(n, rank, world)

    rank_seconds = []

    def timed(fn):
        torch.cuda.synchronize()
        if world > 1 or FORCE_DIST:
            dist.barrier()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        mine_s = time.perf_counter() - t0                     # this rank alone, before it waits for the others
        if world > 1 or FORCE_DIST:
            dist.barrier()
        dt = _max_over_ranks(time.perf_counter() - t0, world, device)
        rank_seconds.append([round(v, 4) for v in _all_ranks(mine_s, world, device)])
        return out, dt
    scoring.COLLECTIVE_LOG = []

    # warm-up on two batches (MIOpen solver search, allocator; two batches so that the loader's worker processes are forked
    # here, outside the timed pass)
    strat.score_pool(mine[:2 * bs], bs)
    # pass 1: through the loader; the uploaded batches are kept for the resident passes
    kept = []

    def loader_pass():
        def gen():
            for b in strat.upload_pool_batches(mine, bs):
                kept.append(b)
                yield b
        return strat.stage1(device_batches=gen())
    rec, dt_loader = timed(loader_pass)
    times = []
    for _ in range(max(1, args.scoring_repeats)):
        rec, dt = timed(lambda: strat.stage1(device_batches=kept))
        times.append(dt)
    med = float(np.median(times))
    assert rec.shape[0] == n and len(strat.bbox_records) == n
    # the same pool at the reference's evaluation batch size (16 frames per batch: tools/cfgs/*/pv_rcnn_active_crb.yaml
    # BATCH_SIZE_PER_GPU / paper supp. B): the frames are re-batched by one untimed loader pass, then resident passes
    ref_bs, at_ref = getattr(args, 'scoring_batch_ref', 0), None
    if ref_bs > 0 and ref_bs != bs:
        kept_ref = list(strat.upload_pool_batches(mine, ref_bs))
        strat.score_device_batches(kept_ref[:2])
        t_ref = [timed(lambda: strat.stage1(device_batches=kept_ref))[1] for _ in range(2)]
        at_ref = {'value': round(n / float(np.median(t_ref)), 3), 'unit': 'frames/s', 'frames_per_batch': ref_bs,
                  'seconds_all': [round(t, 3) for t in t_ref],
                  'note': 'resident passes at the reference\'s evaluation batch size; `value` above uses --scoring-batch frames per '
                          'batch (an eval-mode pass scores every frame on its own: the batch size is the caller\'s loader setting)'}
        del kept_ref
    # kernel table of one scoring batch (event pairs around every C-ABI launch, outside the timed passes)
    ktable = kroof = None
    if not getattr(args, 'no_leg_tables', False):
        with CallProfiler() as cp:
            strat.score_device_batches(kept[:2])
        ktable, kroof = cp.rows(2.0, event_pair_overhead_ms())
    miopen_s = None
    if getattr(args, 'miopen_steps', 0) > 0:
        from pcdet.models.backbones_2d import base_bev_backbone as bev
        keep_flag, bev.WINOGRAD = bev.WINOGRAD, False
        try:
            strat.score_device_batches(kept[:2])
            _, miopen_s = timed(lambda: strat.stage1(device_batches=kept))
        finally:
            bev.WINOGRAD = keep_flag
    # the rest of one selection round at the reference's KITTI budget (SURVEY §8d metric 2: K1 N = 500 frames get gradient
    # embeddings — 16 frames per train-mode pass with per-frame BatchNorm statistics, frames re-read through the loader —
    # k-means++ to K2 N = 300 prototypes with the device restatement of sklearn's seeding, greedy KDE balance to N = 100)
    strat.prototype = 'kmeans++_device'
    if args.stage2_batch > 0:
        strat.stage2_batch = args.stage2_batch
    warm = strat.score_pool(mine[:bs], bs)
    strat.grad_embeddings_batched(mine[:max(bs, strat.stage2_batch)][:strat.stage2_batch], warm[:strat.stage2_batch],
                                  strat.stage2_batch)                              # MIOpen train-mode solver search
    # (the float64 batched GEMM of the device k-means++ loads its library kernels on first use: 0.9 s inside stage 2 in one run,
    # 88 s in a fresh process - warmed like MIOpen's solver search above, on a matrix of the timed shape)
    k1n = min(int(strat.k1 * strat.cfg.ACTIVE_TRAIN.SELECT_NUMS), rec.shape[0])
    if k1n >= 8:
        scoring.kmeans_plusplus_device(torch.randn(k1n, 65536, device=device), 8, random_state=0)
    _, dt_sel = timed(lambda: strat.select_from_records(rec))
    strat.close()
    torch.set_num_threads(host_threads)
    coll, scoring.COLLECTIVE_LOG = scoring.COLLECTIVE_LOG, None
    sel_round = {'stage2_s': round(strat.timings['stage2_s'], 3),
                 'stage2_grad_embeddings_s': round(strat.timings['stage2_embed_s'], 3),
                 'stage3_s': round(strat.timings['stage3_s'], 4), 'stages_2_3_s': round(dt_sel, 3),
                 'round_s_with_resident_stage1': round(med + dt_sel, 3),
                 'K1N': min(strat.k1 * cfg.ACTIVE_TRAIN.SELECT_NUMS, n), 'K2N': min(strat.k2 * cfg.ACTIVE_TRAIN.SELECT_NUMS, n),
                 'N': cfg.ACTIVE_TRAIN.SELECT_NUMS, 'stage2_frames_per_pass': strat.stage2_batch,
                 'clustering': 'kmeans++_device (same picks as sklearn kmeans_plusplus(random_state=0) in the tests)'}
    return {'metric': 'frames/s CRB stage-1 acquisition scoring (PV-RCNN eval, 5 MC-dropout passes, records incl. GT '
                      'statistics, all-gather, per-frame bookkeeping)',
            'value': round(n / med, 3), 'unit': 'frames/s',
            'config': {'workload': 'BASELINE configs[3]: pool of %d synthetic KITTI frames x %d pts, rank-strided shard of %d '
                                   'frames per GPU, batches of %d resident in HBM' % (n, args.points, per, bs),
                       'pool_frames': n, 'frames_per_gpu': per, 'n_gpus': world, 'repeats': len(times),
                       'frames_per_batch': bs},
            'seconds': _pctl(times), 'seconds_all': [round(t, 3) for t in times],
            'through_loader': {'value': round(n / dt_loader, 3), 'unit': 'frames/s', 'seconds': round(dt_loader, 3),
                               'loader_workers': workers,
                               'note': 'one pass, frames generated + collated by the loader workers and uploaded inside '
                                       'the timed region'},
            'at_reference_batch': at_ref,
            'selection_round': sel_round,
            'miopen_convs': None if miopen_s is None else {
                'value': round(n / miopen_s, 3), 'unit': 'frames/s', 'seconds': round(miopen_s, 3),
                'note': 'A/B: one resident pass with MIOpen\'s implicit GEMM for the stride-1 3x3 BEV convolutions (CRB_WINOGRAD=0) '
                        'instead of the hand-written Winograd kernel (BatchNorm folded, bias + ReLU in its epilogue) that `value` runs'},
            'per_rank_seconds': {'loader_pass': rank_seconds[0], 'resident_passes': rank_seconds[1:1 + len(times)],
                                 'note': 'each rank\'s own time for the pass (its scoring + the all-gather it waits in), before the closing barrier'},
            'collectives': [dict(c, seconds=round(c['seconds'], 5)) for c in coll],
            'roofline': kroof, 'kernel_table': ktable, 'kernel_table_unit': 'one batch of %d frames' % bs,
            'record_bytes_per_frame': 4 * strat.layout.stride, 'boxes_kept_total': int(rec[:, 1].sum().item())}


def pvrcnn_bench(args, rank, world, device):
    """BASELINE configs[2]: PV-RCNN fwd+bwd+AdamW, bs=16 per GPU, synthetic KITTI clouds (adds FPS / ball query / grouping /
    RoI-grid pooling HIP kernels to the SECOND path)"""
    from pcdet.datasets import SyntheticDataset
    from pcdet.model_cfgs import pv_rcnn_cfg
    from pcdet.models import build_network
    torch.manual_seed(0)
    model = build_network(pv_rcnn_cfg().MODEL, 3, SyntheticDataset(num_frames=2, n_points=args.points)).to(device)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
    net = model
    if world > 1 or FORCE_DIST:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], gradient_as_bucket_view=True,
                                                        bucket_cap_mb=DDP_BUCKET_MB)
    batches = make_batches(args, rank, device, first=20000)
    for b in batches:
        b['point_frame_counts_host'] = np.diff(b['point_frame_offsets'].cpu().numpy()).tolist()

    ahead = {}                                   # step index -> batch dict whose sparse prologue is already enqueued

    def step(i):
        b = ahead.pop(i, None) or dict(batches[i % len(batches)])
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = net(b)
        if not args.no_sparse_prefetch:          # the next batch's voxel generator + table marks, enqueued before this backward pass
            ahead.clear()
            ahead[i + 1] = model.prefetch_sparse(dict(batches[(i + 1) % len(batches)]))
        loss = ret['loss'].mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()
        return loss
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()         # (after the warm-up: MIOpen's solver search allocates workspaces the steps never see again)
    gc.collect()
    gc.freeze()
    if world > 1 or FORCE_DIST:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.pvrcnn_steps):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1 or FORCE_DIST:
        dist.barrier()
    dt = _max_over_ranks(time.perf_counter() - t0, world, device)
    ktable = kroof = None
    if not getattr(args, 'no_leg_tables', False) and rank == 0:
        with CallProfiler() as cp:                      # two more steps with an event pair around every C-ABI launch (not timed)
            for i in range(2):
                step(args.pvrcnn_steps + i)
        ktable, kroof = cp.rows(2.0, event_pair_overhead_ms())
    out = {'metric': 'frames/s PV-RCNN fwd+bwd+AdamW, KITTI 20k-pt clouds', 'unit': 'frames/s',
           'roofline': kroof, 'kernel_table': ktable, 'kernel_table_unit': 'one training step',
           'kernel_table_note': 'HIP-event pairs around every launch of libcrbhip.so during two extra steps (launches on the side stream '
                                'overlap the main stream: the rows do not add up to the step); Winograd rows: see the top-level kernel_table',
           'value': round(args.batch * world * args.pvrcnn_steps / dt, 3), 'ms_per_step': round(1e3 * dt / args.pvrcnn_steps, 3),
           'steps': args.pvrcnn_steps, 'warmup': 3, 'dtype': 'f32',
           'config': {'workload': 'BASELINE configs[2]: PV-RCNN (VoxelBackBone8x + VSA + PointHeadSimple + PVRCNNHead) on '
                                  'synthetic kitti clouds, %d pts/frame, bs=%d per GPU' % (args.points, args.batch),
                      'global_batch': args.batch * world},
           'peak_mem_GB': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), 'final_loss': round(float(loss.item()), 4)}
    del opt, net, model, batches
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    miopen_db = 'shared'
    if world > 1 and 'MIOPEN_USER_DB_PATH' not in os.environ:
        # N ranks searching MIOpen solvers for the same new shapes would all write ONE user find-db / kernel cache: a db and a
        # cache per rank (set before the first convolution creates MIOpen's handle). The Winograd kernels left MIOpen only the
        # stride-2 3x3 convolution, the 2x2 transposed convolution and the up-sampling GEMMs
        os.environ['MIOPEN_USER_DB_PATH'] = '/tmp/crb_miopen_db_rank%d' % local_rank
        os.environ.setdefault('MIOPEN_CUSTOM_CACHE_DIR', '/tmp/crb_miopen_cache_rank%d' % local_rank)
        os.makedirs(os.environ['MIOPEN_USER_DB_PATH'], exist_ok=True)
        os.makedirs(os.environ['MIOPEN_CUSTOM_CACHE_DIR'], exist_ok=True)
        miopen_db = 'per rank'
    from pcdet.utils.common_utils import effective_cpu_count
    # torch sizes its intra-op pool by os.cpu_count() (128 threads on the 256-thread host) although the pod may use 16 cores:
    # every host-side copy / concat then spins 128 threads on 16 cores, next to the loader workers
    torch.set_num_threads(max(1, min(torch.get_num_threads(), effective_cpu_count() // max(world, 1))))
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % ndev          # one process per GPU; the modulo only matters for the single-GPU gloo dry run
    torch.cuda.set_device(dev_idx)
    device = torch.device('cuda', dev_idx)
    if world > 1 or FORCE_DIST:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('CRB_DIST_BACKEND', 'nccl')          # 'nccl' = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    if os.environ.get('CRB_MIOPEN_FIND', '0') == '1':
        torch.backends.cudnn.benchmark = True       # MIOpen find mode: benchmark the applicable solvers per conv shape
    from crbhip import sparse as sp
    from crbhip import winograd as wino_mod, bnrelu as bn_mod
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
    if world > 1 or FORCE_DIST:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev_idx], gradient_as_bucket_view=True,
                                                        bucket_cap_mb=DDP_BUCKET_MB)
    batches = make_batches(args, rank, device)

    ahead = {}                                   # step index -> batch dict whose sparse prologue is already enqueued

    def step(i, optimizer=True):
        b = ahead.pop(i, None) or dict(batches[i % len(batches)])
        opt.zero_grad(set_to_none=True)
        ret, tb, _ = net(b)
        if not args.no_sparse_prefetch:
            # loader-side work of the NEXT step (voxel generator, marking half of the table plan) goes into the queue before this
            # step's backward pass; its counts reach pinned memory long before the next forward pass asks for them
            # (Detector3DTemplate.prefetch_sparse). Same work per step, no read-back stall in the sparse phase.
            ahead.clear()
            ahead[i + 1] = model.prefetch_sparse(dict(batches[(i + 1) % len(batches)]))
        loss = ret['loss'].mean()
        loss.backward()
        if optimizer:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            opt.step()
        return loss

    rank_step_ms = []          # per timed_steps call: every rank's median device time per step
    rank_wall_ms = []          # ... and every rank's own wall time per step (before the closing barrier)

    def timed_steps(optimizer, profile):
        """-> wall seconds for EXACTLY args.steps steps (barrier + synchronize on both sides, max over ranks) and the
        per-step device-timeline durations (events at the step boundaries on the compute stream, no host sync inside)"""
        torch.cuda.synchronize()
        if world > 1 or FORCE_DIST:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        sp.PROFILE = profile
        wino_mod.PROFILE = profile
        bn_mod.PROFILE = profile
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            loss = step(i, optimizer)
            marks[i + 1].record()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                                  # this rank's own steps, before it waits for the others
        if world > 1 or FORCE_DIST:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rank_wall_ms.append([round(1e3 * v / args.steps, 3) for v in _all_ranks(own, world, device)])
        sp.PROFILE = None
        wino_mod.PROFILE = None
        bn_mod.PROFILE = None
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        rank_step_ms.append([round(v, 3) for v in _all_ranks(float(np.median(per_step)), world, device)])
        return _max_over_ranks(dt, world, device), per_step, loss

    t_warm = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    warm_s = _all_ranks(time.perf_counter() - t_warm, world, device)     # solver search + allocator + table caches, per rank
    # everything alive after warm-up (model, optimizer state, caches, the interpreter's own objects) leaves the garbage collector's
    # generations: a full collection inside a timed step then scans only what the steps themselves created (observed without it:
    # two 8 - 12 ms holes per 8 steps in the kernel trace, step-time p90 50 ms against a 44 ms median)
    gc.collect()
    gc.freeze()
    overhead_ms = event_pair_overhead_ms() if rank == 0 else 0.0
    prof = [] if rank == 0 else None
    torch.cuda.reset_peak_memory_stats()
    dt, per_step, loss = timed_steps(True, prof)
    peak_main = torch.cuda.max_memory_allocated()
    # the same step without grad-clip / AdamW ("one optimizer-less loss.backward() step"); not profiled
    dt_nopt, per_step_nopt, _ = timed_steps(False, None)
    # OPT-IN split-bf16 gather-GEMM (spconv.pytorch.set_arithmetic(model, 'bf16x3')): a few more steps of the same training loop with
    # the contract switched on, only to report its roofline and step time beside the exact-f32 ones; never part of `value`
    prof3, dt3, per_step3 = ([] if rank == 0 else None), None, None
    if args.bf16x3_steps > 0:
        import spconv.pytorch as spconv_mirror
        spconv_mirror.set_arithmetic(model, 'bf16x3')
        try:
            step(0)
            keep = args.steps
            args.steps = args.bf16x3_steps
            dt3, per_step3, loss3 = timed_steps(True, prof3)
            bf16x3_loss = float(loss3.item())
        finally:
            args.steps = keep
            spconv_mirror.set_arithmetic(model, 'f32')
    # A/B: the same training loop for a few more steps with MIOpen's implicit GEMM for the stride-1 3x3 convolutions of the BEV
    # backbone (CRB_WINOGRAD=0) instead of the hand-written Winograd F(2x2,3x3) kernel (crbhip.winograd) that `value` runs
    miopen = None
    if args.miopen_steps > 0:
        from pcdet.models.backbones_2d import base_bev_backbone as bev
        keep_flag, bev.WINOGRAD = bev.WINOGRAD, False
        try:
            step(0)
            step(1)
            keep = args.steps
            args.steps = args.miopen_steps
            dtw, per_step_w, loss_w = timed_steps(True, None)
            miopen = {'frames_per_s': round(args.batch * world * args.miopen_steps / dtw, 3),
                      'ms_per_step': round(1e3 * dtw / args.miopen_steps, 3), 'steps': args.miopen_steps,
                      'ms_per_step_device': _pctl(per_step_w), 'final_loss': round(float(loss_w.item()), 4),
                      'winograd_default': bool(keep_flag),
                      'note': 'A/B, not `value`: same training loop with the 11 stride-1 3x3 BEV convolutions (forward, input '
                              'gradient and weight gradient) on MIOpen\'s implicit GEMM (CRB_WINOGRAD=0) instead of the '
                              'hand-written Winograd F(2x2,3x3) f32 MFMA kernels'}
        finally:
            args.steps = keep
            bev.WINOGRAD = keep_flag
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
                   'parallelism': 'dp%d' % world, 'optimizer': 'grad-clip + fused AdamW in the timed region',
                   'final_loss': round(float(loss.item()), 4)},
        'ms_per_step_device': _pctl(per_step), 'peak_mem_GB': round(peak_main / 2 ** 30, 1),
        'ms_per_step_device_per_rank': rank_step_ms[0], 'ms_per_step_per_rank': rank_wall_ms[0],
        'warmup_seconds_per_rank': [round(v, 3) for v in warm_s], 'miopen_user_db': miopen_db,
        'fwd_bwd_only': {'value': round(frames / dt_nopt, 3), 'unit': 'frames/s',
                         'ms_per_step': round(1e3 * dt_nopt / args.steps, 3), 'ms_per_step_device': _pctl(per_step_nopt),
                         'note': 'same %d steps without grad-clip / optimizer' % args.steps},
    }
    # the kernel tables come out of the event records NOW: the records hold the rulebook tables of every profiled step (device
    # memory: until r04 they stayed alive through the later legs and sat in their peak-memory numbers, 12 GB of the "40.9 GB")
    roof_pack = None
    if rank == 0:
        roof_g, table_g = roofline_from_profile(prof, overhead_ms)
        roof_pack = dominant_roofline(prof, overhead_ms, roof_g, table_g)
        roof3_pack = roofline_from_profile(prof3, overhead_ms, 'bf16x3') if dt3 is not None else None
    prof = prof3 = None
    del opt, net, model, batches
    gc.collect()
    torch.cuda.empty_cache()
    pv = pvrcnn_bench(args, rank, world, device) if (args.pvrcnn_steps > 0 and args.kind == 'kitti') else None
    score = crb_scoring_bench(args, rank, world, device) if (args.scoring_pool > 0 and args.kind == 'kitti') else None
    if rank == 0:
        roof, gather_roof, table = roof_pack
        # the numbers a reader looks for first, once at the head of the line and once at its very end (a log tail keeps the end)
        wino = {k: v for k, v in table.items() if k.startswith('winograd')}
        headline = {
            'second_frames_per_s': out['value'], 'second_ms_per_step': out['ms_per_step'],
            'frames_per_s_per_gpu': round(out['value'] / world, 3),
            # value / (N x the 1-GPU value): needs the N = 1 number of the same code on the same node (--n1-value, e.g. from the
            # N = 1 line of the driver's scaling sweep); null otherwise - the driver computes the curve from the per-N lines itself
            'scaling_efficiency': None if not args.n1_value else round(out['value'] / (world * args.n1_value), 4),
            'crb_scoring_frames_per_s': None if score is None else score['value'],
            'crb_scoring_frames_per_batch': None if score is None else score['config']['frames_per_batch'],
            'crb_scoring_at_reference_batch': None if score is None or not score.get('at_reference_batch') else score['at_reference_batch']['value'],
            'crb_scoring_through_loader': None if score is None else score['through_loader']['value'],
            'crb_stages_2_3_s': None if score is None else score['selection_round']['stages_2_3_s'],
            'pvrcnn_frames_per_s': None if pv is None else pv['value'], 'pvrcnn_ms_per_step': None if pv is None else pv['ms_per_step'],
            'pvrcnn_peak_mem_GB': None if pv is None else pv['peak_mem_GB'],
            'miopen_convs_frames_per_s': None if miopen is None else miopen['frames_per_s'],
            'roofline_frac': None if roof is None else roof['frac'],
            'winograd_mfma_f32_frac': {k: v.get('mfma_f32_frac') for k, v in wino.items()},
            'winograd_avg_us': {k: v.get('avg_us') for k, v in wino.items()},
            'gather_gemm_frac': None if gather_roof is None else gather_roof.get('frac'),
        }
        first = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                     'vs_baseline', 'dtype', 'data', 'config')}
        first['headline'] = headline
        first.update({k: v for k, v in out.items() if k not in first})
        out = first
        out['roofline'] = roof
        if gather_roof is not None:
            out['roofline_gather_gemm'] = gather_roof
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args)
        else:
            out['cpu_baseline'] = None
        out['crb_scoring'] = score
        out['pvrcnn'] = pv
        out['miopen_convs'] = miopen
        out['kernel_table'] = table
        if dt3 is not None:
            roof3, table3 = roof3_pack
            out['roofline_bf16x3'] = roof3
            out['bf16x3'] = {'frames_per_s': round(args.batch * world * args.bf16x3_steps / dt3, 3),
                             'ms_per_step': round(1e3 * dt3 / args.bf16x3_steps, 3), 'steps': args.bf16x3_steps,
                             'ms_per_step_device': _pctl(per_step3), 'final_loss': round(bf16x3_loss, 4),
                             'kernel_table': table3,
                             'note': 'same training loop, gather-GEMM fwd+dgrad of the C>=32 layers under the opt-in '
                                     'split-bf16 contract (wgrad, BEV backbone and heads unchanged, f32)'}
        out['summary'] = headline
        print(json.dumps(out))
    if world > 1 or FORCE_DIST:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
