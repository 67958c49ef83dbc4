#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp37.log
: > $out
for wg in 96 128 160 250; do
echo "== workers $wg" >> $out
GPX_PANEL_WG_WHOLE=$wg GPX_PANEL_WHOLE=4096 timeout -k 10 120 python3 tools/whole_check.py 2048 2560 3072 4096 2>&1 | cut -c1-80 >> $out
done
echo "== off" >> $out
GPX_PANEL_WHOLE=0 timeout -k 10 120 python3 tools/whole_check.py 1536 2048 2560 3072 4096 2>&1 | cut -c1-80 >> $out
cat $out
