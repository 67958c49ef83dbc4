#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp27.log
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_la.py -m gpu -x -q > gpurun_out/r03_gputests14.log 2>&1
tail -6 gpurun_out/r03_gputests14.log >> $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 10 >> $out 2>&1; }
for n in 1024 2048 4096 8192; do
  run "GPX_LEAF_MFMA=0" $n
  run "GPX_LEAF_MFMA=1" $n
done
cat $out
