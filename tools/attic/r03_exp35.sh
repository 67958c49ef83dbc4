#!/bin/bash
# C4: block size of batch members (throughput-bound, the chain hides behind other members)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp35.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/batch_time.py 8192 64 >> $out 2>&1; }
run "GPX_X=0"
run "GPX_NB=2048"
run "GPX_NB=2048 GPX_NB0=2048"
run "GPX_BLOCKS=1024,1024,2048,2048,2048"
run "GPX_X=1"
cat $out
