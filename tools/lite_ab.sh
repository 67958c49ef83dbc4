#!/bin/bash
# Run on the GPU box (through gpurun): 256 thetas at N = 512 / 1024 / 2048 through the round-4
# phases of the lock-step sweep (GPX_SWEEP_LITE=0) and through the dense row panels
# (GPX_SWEEP_LITE=1, fold depth GPX_SWEEP_FOLD), value-only / with gradients, member 0 against
# the single evaluation and eight members against the oracle.
# usage: tools/lite_ab.sh <tag> "<sizes>" "<variants: lite:fold ...>"
set -o pipefail
tag=${1:-lite_ab}; sizes=${2:-"512 1024 2048"}; variants=${3:-"0:-1 1:-1"}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for n in $sizes; do
  for v in $variants; do
    lite=${v%%:*}; fold=${v##*:}
    GPX_SWEEP_LITE=$lite GPX_SWEEP_FOLD=$fold timeout -k 10 300 python3 tools/batch_small.py --b 256 --sizes $n --reps 5 --check > $out/n${n}_l${lite}_f${fold}.json 2> $out/n${n}_l${lite}_f${fold}.err || { echo "FAILED n=$n $v"; tail -5 $out/n${n}_l${lite}_f${fold}.err; exit 1; }
    python3 - $out/n${n}_l${lite}_f${fold}.json $n $v <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('N=%s lite:fold=%-5s value-only %8.0f  with-grad %8.0f evals/s  lZ0 %.12f  m0==single %s/%s  vo==grad lZ %s  oracle err %.1e' % (
    sys.argv[2], sys.argv[3], r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['lZ0'],
    r['member0_equals_single_value_only'], r['member0_equals_single_with_grad'],
    r['value_only_equals_with_grad_lZ'], r['max_rel_err_vs_oracle']))
PY
  done
done 2>&1 | tee $out/summary.txt
