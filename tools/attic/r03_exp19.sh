#!/bin/bash
# round-3: re-check of round-2 knobs under the round-3 defaults (N=16384, one evaluation at a time)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp19.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "BASE=1" 16384
run "GPX_GEMM_SMALL_BELOW=200" 16384
run "GPX_GEMM_SMALL_BELOW=800" 16384
run "GPX_GEMM_SMALL_BELOW=1600" 16384
run "GPX_LDPAD=0" 16384
run "GPX_LDPAD=16" 16384
run "GPX_LDPAD=64" 16384
run "GPX_LDPAD=160" 16384
run "BASE=1" 16384
run "GPX_PANEL=512" 16384
run "GPX_NB=1024" 16384
run "GPX_TRACE_ROWS=8" 16384
run "GPX_TRACE_ROWS=32" 16384
run "GPX_SWIZZLE=1" 16384
run "GPX_ORD_R12=0" 16384
run "GPX_ORD_T=0" 16384
run "BASE=1" 16384
cat $out
