#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: per kernel (name, grid) the sum
of each counter, plus the derived figures the profiles quote:

  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)
      GRBM_GUI_ACTIVE is summed over the 8 XCDs (so / 8 = cycles the kernel ran);
      1024 = 256 CUs x 4 SIMDs, each with one matrix pipe whose busy cycles
      SQ_VALU_MFMA_BUSY_CYCLES adds up. 1.0 = every matrix pipe busy every cycle.
  VALU busy        = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024)
      SQ_ACTIVE_INST_* count quad-cycles per SIMD (MI355X_MICROARCH.md, cycle table).
  effective clock  = GRBM_GUI_ACTIVE / 8 / kernel wall time is not available here
      (no timestamps in a PMC pass): see the kernel-trace summary of the same run.
  HBM bytes        = 2 x FETCH_SIZE KiB (gfx950 counts 128-B requests as 64 B) and
                     WRITE_SIZE KiB as is.

usage: pmc_summary.py <counter_collection.csv> [min_grid] [name filter regex]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ming = int(sys.argv[2]) if len(sys.argv) > 2 else 0
filt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    gs = int(r['Grid_Size']) if 'Grid_Size' in r else 0
    if gs < ming or (filt and not filt.search(name)):
        continue
    key = (name, gs)
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(key, r['Counter_Name'])] += 1
for key in sorted(agg):
    print(key[0], 'grid', key[1])
    c = agg[key]
    for name, v in sorted(c.items()):
        n = cnt[(key, name)]
        print('    %-28s %16.4g  (per dispatch %12.4g, %d dispatches)' % (name, v, v / n, n))
    if c.get('GRBM_GUI_ACTIVE'):
        simd_cycles = c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
            print('    => MFMA utilisation %.3f  (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 '
                  'x 1024 SIMDs))' % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles))
        if 'SQ_ACTIVE_INST_VALU' in c:
            print('    => VALU busy        %.3f  (4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE/8 '
                  'x 1024 SIMDs))' % (4.0 * c['SQ_ACTIVE_INST_VALU'] / simd_cycles))
        if 'SQ_WAVE_CYCLES' in c:
            print('    => waves per SIMD   %.2f  (4 x SQ_WAVE_CYCLES / SIMD-cycles)'
                  % (4.0 * c['SQ_WAVE_CYCLES'] / simd_cycles))
    if 'SQ_WAVES' in c and 'SQ_INSTS_VALU' in c and c['SQ_WAVES']:
        print('    => VALU instructions per wave %.0f' % (c['SQ_INSTS_VALU'] / c['SQ_WAVES']))
    if 'FETCH_SIZE' in c:
        print('    => HBM read  %.4g GB per dispatch (2 x FETCH_SIZE KiB, gfx950 correction)'
              % (2 * 1024.0 * c['FETCH_SIZE'] / cnt[(key, 'FETCH_SIZE')] * 1e-9))
    if 'WRITE_SIZE' in c:
        print('    => HBM write %.4g GB per dispatch (WRITE_SIZE KiB)'
              % (1024.0 * c['WRITE_SIZE'] / cnt[(key, 'WRITE_SIZE')] * 1e-9))
