#!/usr/bin/env python3
"""Compact timeline of a rocprofv3 kernel trace: per launch start offset, duration, queue,
workgroups and a short kernel name, for the launches between two dispatch ids.
usage: python tools/trace_timeline.py <kernel_trace.csv> [first_dispatch [last_dispatch]]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
sel = [r for r in rows if lo <= int(r['Dispatch_Id']) <= hi]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel:
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')
    name = name.replace('Geo<', 'G<').replace(', true>', ',D>')
    wg = 1
    for ax in 'XYZ':
        wg *= max(1, int(r['Grid_Size_' + ax]) // max(1, int(r['Workgroup_Size_' + ax])))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%5s q%-2s %9.1f us  +%8.1f us  wg %6d  %s' % (r['Dispatch_Id'], r['Queue_Id'],
                                                         (s - t0) / 1e3, (e - s) / 1e3, wg, name[:70]))
