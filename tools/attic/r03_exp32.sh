#!/bin/bash
# C4 (64 theta x N=8192): panel workers and batch depth
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp32.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/batch_time.py 8192 64 >> $out 2>&1; }
run "GPX_X=0"
run "GPX_PANEL_WG=8"
run "GPX_PANEL_WG=16"
run "GPX_PANEL_WG=24"
run "GPX_PANEL_WG=48"
run "GPX_BATCH_INFLIGHT=4"
run "GPX_BATCH_INFLIGHT=4 GPX_PANEL_WG=16"
run "GPX_X=1"
cat $out
