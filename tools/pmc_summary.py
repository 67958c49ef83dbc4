#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel (name, grid)
sum of each counter. usage: pmc_summary.py <counter_collection.csv> [min_grid]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ming = int(sys.argv[2]) if len(sys.argv) > 2 else 0
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:34]
    gs = int(r['Grid_Size']) if 'Grid_Size' in r else 0
    if gs < ming: continue
    key = (name, gs)
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(key, r['Counter_Name'])] += 1
for key in sorted(agg):
    print(key[0], 'grid', key[1])
    for c, v in sorted(agg[key].items()):
        n = cnt[(key, c)]
        print('    %-28s %16.4g  (per dispatch %12.4g, %d dispatches)' % (c, v, v / n, n))
