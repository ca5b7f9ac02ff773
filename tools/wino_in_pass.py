#!/usr/bin/env python
"""From a rocprofv3 kernel trace of tools/prof_scoring_resident.py: start offsets (relative to the pass's vox_insert) and durations of
the Winograd launches, of the FPS kernel and of the CU-reservation marks in the last complete pass.  usage: python tools/wino_in_pass.py <kernel_trace.csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'vox_insert' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name']
    if 'winograd2_kernel' in n or 'fps_kernel' in n or 'cu_busy' in n:
        print('%8.1f us  +%7.1f us  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, n[:60]))
print('pass length %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
