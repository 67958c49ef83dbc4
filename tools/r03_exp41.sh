#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp41.log
: > $out
for v in 0 1; do
echo "== GPX_PANEL_RHS=$v" >> $out
GPX_PANEL_RHS=$v timeout -k 10 200 python3 tools/whole_check.py 1100 1536 2048 3001 3072 4096 2>&1 | cut -c1-200 >> $out
done
cat $out
