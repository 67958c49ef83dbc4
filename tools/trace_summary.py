#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: per kernel name and grid, calls and
time per evaluation. usage: trace_summary.py <kernel_trace.csv> <n_evals>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nev = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(float)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    gx = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    key = (name, gx, int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key][0] += 1
    agg[key][1] += d
    tot[name] += d
for name, t in sorted(tot.items(), key=lambda x: -x[1]):
    print('%-40s %10.3f ms/eval' % (name, t / nev / 1e3))
print()
for k, v in sorted(agg.items()):
    if v[1] / nev > 200:
        print('%-34s grid=(%d,%d,%d) calls/eval %6.1f total/eval %8.2f ms avg %9.1f us'
              % (k[0], k[1], k[2], k[3], v[0] / nev, v[1] / nev / 1e3, v[1] / v[0]))
