#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace of ONE factorisation / evaluation (the
last kbuild launch to the end): busy time per HW queue and the long kernels in start
order. usage: trace_queues.py <kernel_trace.csv> [min_us]"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
kb = [i for i, r in enumerate(rows) if 'kbuild' in r['Kernel_Name']]
sel = rows[kb[-1]:]
T0 = int(sel[0]['Start_Timestamp']); end = max(int(r['End_Timestamp']) for r in sel)
print('span ms %.2f' % ((end - T0) / 1e6))
qs = collections.defaultdict(list)
for r in sel: qs[r['Queue_Id']].append(r)
for q, rs in sorted(qs.items()):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print('queue %s: %4d kernels, busy %.2f ms, first start %.2f, last end %.2f' % (
        q, len(rs), busy / 1e6, (int(rs[0]['Start_Timestamp']) - T0) / 1e6,
        (max(int(r['End_Timestamp']) for r in rs) - T0) / 1e6))
for r in sel:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > minus:
        nm = r['Kernel_Name'].split('(')[0].replace('void ', '')[:36]
        print('%8.2f %8.2f q=%s %-36s wg=%d' % ((int(r['Start_Timestamp']) - T0) / 1e6, d / 1e3,
              r['Queue_Id'], nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
