#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp23.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/batch_time.py 16384 9 >> $out 2>&1; }
run "BASE=1"
run "GPX_TILE_XCD=1"
run "GPX_PANEL_WG=16"
run "GPX_PANEL_WG=24"
run "GPX_PANEL_WG=48"
run "GPX_OVERLAP_NOSPLIT=0"
run "GPX_AUX=0"
run "BASE=1"
cat $out
