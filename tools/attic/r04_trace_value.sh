#!/bin/bash
# kernel trace of value-only evaluations one at a time at N = $1 -> timeline of the last one
set -o pipefail
n=${1:-4096}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_trace_value_$n; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -- python3 tools/run_value.py $n 6 > $out/run.log 2>&1 || exit 1
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_timeline.py $f | tail -14
python3 tools/seq_time.py $n 8
