#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp14.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "BASE=1" 16384
run "GPX_NB0=512" 16384
run "GPX_NB0=2048" 16384
run "GPX_BLOCKS=512,1536,2048" 16384
run "GPX_BLOCKS=1024,1024,2048" 16384
run "BASE=1" 16384
run "GPX_NB0=512" 8192
run "BASE=1" 8192
cat $out
