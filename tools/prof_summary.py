#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state steps only.

MIOpen's find phase (naive_conv_*, GEMM trials) pollutes whole-process --stats; a training step of this repo launches
the voxelizer's `vox_insert` kernel exactly once, so the trace is cut at vox_insert launches and only K complete steps
near the end of the trace are aggregated.  Usage: prof_summary.py <kernel_trace.csv> <K> > summary.csv
PROF_MARKER=<kernel substring> picks another once-per-step kernel; PROF_SPLIT_GRID=<kernel substring> lists the matching
kernel once per launch geometry (the subm and the strided layers run the same gather-GEMM instance);
PROF_LIST=<substr,substr> lists every matching launch of the last complete step (start offset, duration, grid)."""
import csv
import sys
from collections import defaultdict


def main(path, K, marker='vox_insert'):
    import os
    marker = os.environ.get('PROF_MARKER', marker)
    rows = list(csv.DictReader(open(path)))
    key_s = 'Start_Timestamp' if 'Start_Timestamp' in rows[0] else 'Start'
    key_e = 'End_Timestamp' if 'End_Timestamp' in rows[0] else 'End'
    key_n = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
    rows.sort(key=lambda r: int(r[key_s]))
    marks = [i for i, r in enumerate(rows) if marker in r[key_n]]
    if len(marks) <= K + 1:
        raise SystemExit('not enough steps in trace: %d markers' % len(marks))
    # K COMPLETE steps: from the (K+1)-th last marker up to (not including) the last one — what follows the last step of a
    # process (final synchronisation, result read-backs, bookkeeping kernels) is not part of any step (until r02 the window
    # ran to the end of the trace and charged one ~5 ms post-run gap to the steps)
    sel = rows[marks[-K - 1]:marks[-1]]
    span_ns = int(sel[-1][key_e]) - int(sel[0][key_s])
    agg = defaultdict(lambda: [0, 0])
    split = os.environ.get('PROF_SPLIT_GRID', '')          # e.g. "sparse_conv_fwd2": one line per launch geometry
    key_g = next((k for k in ('Grid_Size_X', 'Grid_Size', 'Grid_Size_x') if k in rows[0]), None)
    for r in sel:
        d = int(r[key_e]) - int(r[key_s])
        name = r[key_n]
        if split and key_g and split in name:
            name = '[grid_x=%s] %s' % (r[key_g], name)
        a = agg[name]
        a[0] += 1
        a[1] += d
    busy = sum(v[1] for v in agg.values())
    # union of the kernel intervals (streams overlap) -> time with NO kernel running, and the largest gaps
    iv = sorted((int(r[key_s]), int(r[key_e]), r[key_n]) for r in sel)
    union, gaps, cur_e, cur_name = 0, [], iv[0][0], ''
    for s0, e0, nm in iv:
        if s0 > cur_e:
            gaps.append((s0 - cur_e, cur_name, nm))
            union += 0
            cur_s = s0
        union += max(0, e0 - max(s0, cur_e))
        if e0 > cur_e:
            cur_e, cur_name = e0, nm
    idle = span_ns - union
    w = csv.writer(sys.stdout)
    w.writerow(['# steps', K, 'wall_ms_per_step', '%.3f' % (span_ns / 1e6 / K), 'kernel_busy_ms_per_step',
                '%.3f' % (busy / 1e6 / K), 'gpu_idle_ms_per_step (no kernel on any stream)', '%.3f' % (idle / 1e6 / K),
                'gaps_over_20us_per_step', '%.1f' % (sum(1 for g in gaps if g[0] > 20000) / K)])
    if os.environ.get('PROF_GAPS'):
        for g in sorted(gaps, reverse=True)[:int(os.environ['PROF_GAPS'])]:
            w.writerow(['# gap_us', '%.1f' % (g[0] / 1e3), 'after', g[1][:70], 'before', g[2][:70]])
    if os.environ.get('PROF_LIST'):
        # every launch of the LAST complete step whose name holds one of the comma-separated substrings, in time order:
        # start offset inside the step, duration, grid, name (per-launch view of kernels that run once per table / level)
        pats = [x for x in os.environ['PROF_LIST'].split(',') if x]
        last = rows[marks[-2]:marks[-1]]
        t0 = int(last[0][key_s])
        for r in last:
            if any(x in r[key_n] for x in pats):
                w.writerow(['# launch', '%.1f' % ((int(r[key_s]) - t0) / 1e3), 'dur_us', '%.1f' % ((int(r[key_e]) - int(r[key_s])) / 1e3),
                            'grid', r.get(key_g, ''), r[key_n][:60]])
    w.writerow(['kernel', 'calls_per_step', 'avg_us', 'ms_per_step', 'pct_of_busy'])
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name[:160], '%.2f' % (n / K), '%.2f' % (t / n / 1e3), '%.4f' % (t / 1e6 / K), '%.2f' % (100.0 * t / busy)])


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
