#!/bin/bash
# round-3: tile rows of the tall inverse-column product over 1 / 2 / 4 XCDs: time and HBM reads
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_exp15.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
for g in 0 4 2 1 0 4 2; do run "GPX_TILE_ROWGRP=$g" 16384; done
for g in 0 2 4; do
  d=gpurun_out/r03_exp15_fetch$g
  GPX_TILE_ROWGRP=$g rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 tools/run_eval.py 16384 1 > $d.log 2>&1 || exit 1
  f=$(ls $d/*/*counter_collection.csv | head -1)
  python3 - "$f" $g >> $out <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE': tot += float(r['Counter_Value'])
print('rowgrp %s: HBM reads per evaluation %.1f GB (2 x FETCH_SIZE KiB)' % (sys.argv[2], tot * 2 * 1024 / 1e9))
PY
done
cat $out
