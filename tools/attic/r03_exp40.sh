#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/r03_tr41
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_tr41 -- python3 tools/run_value.py 4096 8 > gpurun_out/r03_tr41.log 2>&1 || { echo failed; tail -3 gpurun_out/r03_tr41.log; exit 1; }
f=$(find gpurun_out/r03_tr41 -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] || { echo "no trace"; exit 1; }
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last evaluation: from the last kbuild_kernel<double pair backwards
idx = [i for i, r in enumerate(rows) if 'kbuild_kernel<double' in r['Kernel_Name']]
start = idx[-2]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    print('%8.1f us +%7.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:60]))
PY
