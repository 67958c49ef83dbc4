#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp37.log
: > $out
for kb in 8 10 12 16; do
echo "== kbatch $kb" >> $out
GPX_PANEL_KBATCH=$kb timeout -k 10 120 python3 tools/whole_check.py 1536 2048 3072 4096 2>&1 | cut -c1-88 >> $out
done
cat $out
