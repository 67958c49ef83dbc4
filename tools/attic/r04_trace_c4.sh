#!/bin/bash
# kernel trace of C4 batches (64 thetas x N = 8192): value-only, then with gradients
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_trace_c4; mkdir -p $out
for g in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/g$g -- python3 tools/run_batch.py 8192 64 2 8 $g > $out/run$g.log 2>&1 || exit 1
  f=$(ls $out/g$g/*/*kernel_stats.csv | head -1)
  echo "== grad=$g"; head -14 "$f" | cut -c1-150
done
