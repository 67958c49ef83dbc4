#!/bin/bash
# Run on the GPU box (through gpurun): 256 thetas at small N under sets of environment switches.
# usage: tools/env_ab.sh <tag> "<sizes>" "<variant> <variant> ..."   (variant: A=1,B=2 or "-")
set -o pipefail
tag=${1:-env_ab}; sizes=${2:-"512 1024 2048"}; variants=${3:-"-"}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do
  for v in $variants; do
    envs=$(echo "$v" | tr ',' ' '); [ "$v" = "-" ] && envs="GPX_NOOP=1"
    f=$out/n${n}_$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
    env $envs timeout -k 10 300 python3 tools/batch_small.py --b 256 --sizes $n --reps 5 --check > $f.json 2> $f.err || { echo "FAILED n=$n $v"; tail -5 $f.err; exit 1; }
    python3 - $f.json $n "$v" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('N=%s %-40s value-only %8.0f  with-grad %8.0f evals/s  m0==single %s/%s  oracle err %.1e' % (
    sys.argv[2], sys.argv[3], r['value_only_evals_per_s'], r['with_grad_evals_per_s'],
    r['member0_equals_single_value_only'], r['member0_equals_single_with_grad'], r['max_rel_err_vs_oracle']))
PY
  done
done 2>&1 | tee $out/summary.txt
