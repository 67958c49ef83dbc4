#!/bin/bash
# round-3: kernel timeline of one evaluation at N=4096 and N=8192 (every kernel, per queue)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in 4096 8192; do
  out=gpurun_out/r03_exp12_$n; mkdir -p $out
  rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/run_eval.py $n 4 > $out/trace.log 2>&1 || exit 1
  f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
  python3 tools/trace_queues.py $f 0 > $out/queues.txt
done
head -80 gpurun_out/r03_exp12_4096/queues.txt
