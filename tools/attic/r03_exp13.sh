#!/bin/bash
# round-3: wide panel launches (GPX_PANEL_WIDE)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp13.log
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_la.py tests/test_gpu_gp.py -m gpu -x -q > gpurun_out/r03_gputests6.log 2>&1
tail -5 gpurun_out/r03_gputests6.log >> $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
for n in 4096 8192; do
  run "GPX_PANEL_WIDE=0" $n
  run "GPX_PANEL_WIDE=1" $n
  run "GPX_PANEL_WIDE=1 GPX_PANEL_WG_WIDE=64" $n
  run "GPX_PANEL_WIDE=1 GPX_PANEL_WG_WIDE=96" $n
done
run "GPX_PANEL_WIDE=0" 2048
run "GPX_PANEL_WIDE=1" 2048
run "GPX_PANEL_WIDE=1" 16384
run "GPX_PANEL_WIDE=2" 16384
run "GPX_PANEL_WIDE=2 GPX_PANEL_WG_WIDE=32" 16384
cat $out
