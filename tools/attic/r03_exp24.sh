#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp24.log
: > $out
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
c4 "BASE=1"
c4 "GPX_RESERVE_CUS=32 GPX_BATCH_LOOKAHEAD=1"
c4 "GPX_RESERVE_CUS=16 GPX_BATCH_LOOKAHEAD=1"
c4 "GPX_RESERVE_CUS=32"
c4 "GPX_RESERVE_CUS=32 GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=2"
cat $out
