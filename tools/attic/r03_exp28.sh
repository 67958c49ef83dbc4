#!/bin/bash
# XS task with the diagonal update interleaved: LA tests, timings, panel trace
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp28.log
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_la.py -m gpu -x -q > gpurun_out/r03_gputests20.log 2>&1
tail -6 gpurun_out/r03_gputests20.log >> $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 10 >> $out 2>&1; }
for n in 1024 2048 4096 8192 16384; do
  run "GPX_X=0" $n
done
GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 timeout -k 10 120 python3 tools/panel_dbg.py 1024 1024 > gpurun_out/r03_ptrace_10.log 2>&1
python3 tools/panel_trace_summary.py gpurun_out/r03_ptrace_10.log | tail -30 >> $out
cat $out
