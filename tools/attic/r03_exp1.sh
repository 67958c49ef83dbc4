#!/bin/bash
# round-3 A/B: panel workgroups at N=16384 / 8192 (fewer CUs held by the latency-bound chain)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp1.log
: > $out
run() { TAG="$1" env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "BASE=1" 16384
for wg in 8 16 24 32 48; do run "GPX_PANEL_WG=$wg" 16384; done
run "BASE=1" 8192
for wg in 16 32 128; do run "GPX_PANEL_WG=$wg" 8192; done
run "BASE=1" 4096
cat $out
