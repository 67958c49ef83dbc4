#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp22.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/batch_time.py $2 $3 >> $out 2>&1; }
for n in 16384 12288 10240; do
  run "BASE=1" $n 9
  run "GPX_BATCH_LOOKAHEAD=1" $n 9
  run "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=4 GPU_MAX_HW_QUEUES=8" $n 12
  run "GPX_BATCH_INFLIGHT=4 GPU_MAX_HW_QUEUES=8" $n 12
done
cat $out
