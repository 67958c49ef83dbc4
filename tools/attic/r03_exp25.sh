#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp25.log
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_la.py tests/test_gpu_gp.py -m gpu -x -q > gpurun_out/r03_gputests13.log 2>&1
tail -4 gpurun_out/r03_gputests13.log >> $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 8 >> $out 2>&1; }
for n in 2048 4096 8192 16384; do
  run "GPX_INVCOL_EARLY=0" $n
  run "GPX_INVCOL_EARLY=1" $n
done
run "GPX_INVCOL_EARLY=0" 16384
run "GPX_INVCOL_EARLY=1" 16384
TAG=early0 GPX_INVCOL_EARLY=0 python3 tools/batch_time.py 16384 9 >> $out 2>&1
TAG=early1 GPX_INVCOL_EARLY=1 python3 tools/batch_time.py 16384 9 >> $out 2>&1
cat $out
