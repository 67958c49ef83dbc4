#!/bin/bash
# Run on the GPU box (through gpurun): kernel traces of 256 thetas at N = 512, 1024, 2048
# (tools/batch_small.py --reps 1) with the full launch timeline of each run kept beside the
# stats -> gpurun_out/<tag>/ ; usage: tools/prof_small_timeline.sh <tag> [sizes]
set -o pipefail
tag=${1:-small_tl}; sizes=${2:-"512 1024 2048"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$n -- python3 tools/batch_small.py --b 256 --sizes $n --reps ${REPS:-1} > $out/run_$n.log 2>&1 || exit 1
  f=$(ls $out/trace_$n/*/*kernel_trace.csv | head -1)
  python3 tools/trace_timeline.py "$f" > $out/timeline_$n.txt
  cp $(ls $out/trace_$n/*/*kernel_stats.csv | head -1) $out/stats_$n.csv
  rm -rf $out/trace_$n
  python3 tools/batch_small.py --b 256 --sizes $n --reps 5 > $out/plain_$n.json 2>&1 || exit 1
  echo "N=$n done"
done
