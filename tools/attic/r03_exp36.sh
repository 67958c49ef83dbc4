#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp36.log
: > $out
run() { TAG="$1" timeout -k 10 200 env $1 python3 tools/seq_time.py $2 8 >> $out 2>&1; }
for n in 16384 8192; do
run "GPX_X=0" $n
run "GPX_PANEL_WG=64" $n
run "GPX_PANEL_WG=128" $n
done
cat $out
