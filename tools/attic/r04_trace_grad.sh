#!/bin/bash
# kernel trace of log-lik+grad evaluations one at a time at N = $1 -> timeline of the last one
set -o pipefail
n=${1:-4096}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_trace_grad_$n; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/run_eval.py $n 8 > $out/run.log 2>&1 || exit 1
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_timeline.py $f > $out/timeline.txt
tail -${2:-30} $out/timeline.txt
