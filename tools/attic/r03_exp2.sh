#!/bin/bash
# round-3: per-launch efficiency of the tile engine with everything on ONE stream
# (GPX_LOOKAHEAD=0): what each shape of the real sequence costs when it runs alone
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03_exp2; mkdir -p $out
export GPX_LOOKAHEAD=0
export GPX_GEMM_LOG=$out/gemmlog_la0.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_la0 -- python3 tools/run_eval.py 16384 3 > $out/trace_la0.log 2>&1 || exit 1
unset GPX_GEMM_LOG
f=$(ls $out/trace_la0/*/*kernel_trace.csv | head -1)
python3 tools/gemm_trace_join.py $out/gemmlog_la0.txt $f 50 > $out/launches_la0.txt
tail -40 $out/launches_la0.txt
