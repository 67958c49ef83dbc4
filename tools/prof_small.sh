#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel trace of the batched-theta path at a
# small size -> gpurun_out/<tag>/ ; usage: tools/prof_small.sh <tag> <N> [B]
set -o pipefail
tag=${1:-r04_small}; n=${2:-1024}; b=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/batch_small.py --b $b --sizes $n --reps 1 > $out/run.log 2>&1 || exit 1
f=$(ls $out/trace/*/*kernel_stats.csv | head -1)
head -25 "$f" | cut -c1-200
