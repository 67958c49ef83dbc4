#!/bin/bash
# Run on the GPU box (through gpurun): 256 thetas at small N for group shapes
# (GPX_GROUP_MEMBERS x GPX_GROUP_INFLIGHT) with the dense row panels on or off.
# usage: tools/group_shape_ab.sh <tag> "<sizes>" "<variants: lite:members:inflight ...>" [B]
set -o pipefail
tag=${1:-shape_ab}; sizes=${2:-"512 1024 2048"}; variants=${3:-"1:0:0"}; B=${4:-256}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do
  for v in $variants; do
    IFS=: read lite mem inf <<< "$v"
    envs="GPX_SWEEP_LITE=$lite"
    [ "$mem" != "0" ] && envs="$envs GPX_GROUP_MEMBERS=$mem"
    [ "$inf" != "0" ] && envs="$envs GPX_GROUP_INFLIGHT=$inf"
    f=$out/n${n}_${lite}_${mem}_${inf}
    env $envs timeout -k 10 300 python3 tools/batch_small.py --b $B --sizes $n --reps 5 > $f.json 2> $f.err || { echo "FAILED n=$n $v"; tail -5 $f.err; exit 1; }
    python3 - $f.json $n $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('N=%s lite:members:inflight=%-9s value-only %8.0f  with-grad %8.0f evals/s  m0==single %s/%s' % (
    sys.argv[2], sys.argv[3], r['value_only_evals_per_s'], r['with_grad_evals_per_s'],
    r['member0_equals_single_value_only'], r['member0_equals_single_with_grad']))
PY
  done
done 2>&1 | tee $out/summary.txt
