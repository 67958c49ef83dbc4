#!/bin/bash
# kernel trace of two batched steps of the headline config -> per-kernel totals of the last step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_trace_bench; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline > $out/run.json 2> $out/run.err || exit 1
f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_timeline.py $f > $out/timeline.txt
wc -l $out/timeline.txt
