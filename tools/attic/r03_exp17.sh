#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp17.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "BASE=1" 16384
run "GPX_GEMM_NOSPLIT=1" 16384
run "GPX_AUX=0" 16384
run "GPX_FASTCHAIN=0" 16384
run "GPX_TILE_XCD=0" 16384
run "BASE=1" 16384
run "GPX_LOOKAHEAD=0" 16384
cat $out
