#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state steps only.

MIOpen's find phase (naive_conv_*, GEMM trials) pollutes whole-process --stats; a training step of this repo launches
the voxelizer's `vox_insert` kernel exactly once, so the trace is cut at the (last-K)-th vox_insert and only the
K last steps are aggregated.  Usage: prof_summary.py <kernel_trace.csv> <K> > summary.csv
PROF_MARKER=<kernel substring> picks another once-per-step kernel; PROF_SPLIT_GRID=<kernel substring> lists the matching
kernel once per launch geometry (the subm and the strided layers run the same gather-GEMM instance)."""
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
    if len(marks) <= K:
        raise SystemExit('not enough steps in trace: %d markers' % len(marks))
    first = marks[-K]
    sel = rows[first:]
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
    w = csv.writer(sys.stdout)
    w.writerow(['# steps', K, 'wall_ms_per_step', '%.3f' % (span_ns / 1e6 / K), 'kernel_busy_ms_per_step',
                '%.3f' % (busy / 1e6 / K)])
    w.writerow(['kernel', 'calls_per_step', 'avg_us', 'ms_per_step', 'pct_of_busy'])
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name[:160], '%.2f' % (n / K), '%.2f' % (t / n / 1e3), '%.4f' % (t / 1e6 / K), '%.2f' % (100.0 * t / busy)])


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
