#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp37.log
: > $out
for big in 1 0; do for wg in 160 250; do
echo "== big $big workers $wg" >> $out
GPX_PANEL_BIG_TASKS=$big GPX_PANEL_WG_WHOLE=$wg GPX_PANEL_WHOLE=4096 timeout -k 10 120 python3 tools/whole_check.py 2304 3072 3584 4096 2>&1 | cut -c1-88 >> $out
done; done
echo "== kbatch 8 big" >> $out
GPX_PANEL_KBATCH=8 GPX_PANEL_WHOLE=4096 timeout -k 10 120 python3 tools/whole_check.py 3072 4096 2>&1 | cut -c1-88 >> $out
echo "== blocked" >> $out
GPX_PANEL_WHOLE=0 timeout -k 10 120 python3 tools/whole_check.py 2304 3072 3584 4096 2>&1 | cut -c1-88 >> $out
cat $out
