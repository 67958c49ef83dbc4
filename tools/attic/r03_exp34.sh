#!/bin/bash
# headline batch: members per call and depth
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp34.log
: > $out
run() { TAG="$1 B=$2" timeout -k 10 300 env $1 python3 tools/batch_time.py 16384 $2 >> $out 2>&1; }
run "GPX_X=0" 6
run "GPX_X=0" 12
run "GPX_X=0" 24
run "GPX_BATCH_INFLIGHT=2" 6
run "GPX_BATCH_INFLIGHT=2" 12
run "GPX_BATCH_INFLIGHT=2" 24
run "GPX_BATCH_LOOKAHEAD=0" 12
cat $out
