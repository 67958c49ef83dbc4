#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp41.log
: > $out
for wg in 96 128 160 200 250; do
echo "== workers $wg" >> $out
GPX_PANEL_WG_WHOLE=$wg timeout -k 10 200 python3 tools/whole_check.py 1536 2048 3072 4096 2>&1 | cut -c1-88 >> $out
done
cat $out
