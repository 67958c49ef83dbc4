#!/bin/bash
# Run on the GPU box (through gpurun): 256 thetas at small N through the lock-step sweep
# (GPX_SOLO_MAX_NP=0) and through solo launches (one workgroup per member, GPX_SOLO_MAX_NP=<np>),
# value-only / with gradients, each with member 0 against the single evaluation and eight
# members against the oracle. usage: tools/solo_ab.sh <tag> "<sizes>" [max_np]
set -o pipefail
tag=${1:-solo_ab}; sizes=${2:-"256 384 512 768 1024"}; maxnp=${3:-1024}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do
  for mode in 0 $maxnp; do
    GPX_SOLO_MAX_NP=$mode timeout -k 10 300 python3 tools/batch_small.py --b 256 --sizes $n --reps 5 --check > $out/n${n}_solo${mode}.json 2> $out/n${n}_solo${mode}.err || { echo "FAILED n=$n mode=$mode"; tail -5 $out/n${n}_solo${mode}.err; exit 1; }
    python3 - $out/n${n}_solo${mode}.json $n $mode <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('N=%s solo_max_np=%-5s value-only %8.0f  with-grad %8.0f evals/s  lZ0 %.12f  m0==single %s/%s  vo==grad lZ %s  oracle err %.1e' % (
    sys.argv[2], sys.argv[3], r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['lZ0'],
    r['member0_equals_single_value_only'], r['member0_equals_single_with_grad'],
    r['value_only_equals_with_grad_lZ'], r['max_rel_err_vs_oracle']))
PY
  done
done 2>&1 | tee $out/summary.txt
