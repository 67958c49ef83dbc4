#!/bin/bash
# trace-gradient kernel: kernel durations from a kernel trace (optionally parity tests first)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r03_exp30.log
: > $out
if [ "$1" = test ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_gp.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/r03_gputests21.log 2>&1
  tail -4 gpurun_out/r03_gputests21.log >> $out
fi
rm -rf gpurun_out/r03_tr30
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_tr30 -- python3 tools/run_hbm.py 3 > gpurun_out/r03_tr30.log 2>&1 || { echo "rocprof failed" >> $out; cat $out; exit 1; }
f=$(find gpurun_out/r03_tr30 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] || { echo "no stats file" >> $out; cat $out; exit 1; }
grep -E "trace_grad|xscale" "$f" | awk -F'","' '{printf "%s calls=%s avg_ns=%s\n", substr($1,2,70), $2, $4}' >> $out
timeout -k 10 200 python3 tools/seq_time.py 16384 6 >> $out 2>&1
cat $out
