#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp42.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/seq_time.py 16384 6 >> $out 2>&1; }
run "GPX_X=0"
run "GPX_BLOCKS=1024,2048,2048,2048,2048,2048,1024,1024,1024,1024,1024"
run "GPX_BLOCKS=1024,2048,2048,2048,2048,2048,2048,1024,1024,1024"
run "GPX_BLOCKS=1024,2048,2048,2048,2048,1024,1024,1024,1024,1024,1024,1024"
run "GPX_BLOCKS=1024,3072,3072,3072,2048,1024,1024,1024,1024"
cat $out
