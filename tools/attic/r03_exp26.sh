#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp26.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 8 >> $out 2>&1; }
for n in 6144 8192 12288 16384; do
  run "GPX_GEMM_NOBALANCE=1" $n
  run "GPX_GEMM_NOBALANCE=0" $n
done
c4() {
  echo "== c4 $1" >> $out
  timeout -k 10 300 env $1 python3 tools/bench_configs.py c4 2>> $out | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value-only %.1f evals/s  grad %.1f evals/s  one %.2f ms' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['one_eval_with_grad_ms']))
" >> $out
}
c4 "GPX_GEMM_NOBALANCE=1"
c4 "GPX_GEMM_NOBALANCE=0"
cat $out
