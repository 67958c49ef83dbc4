#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp21.log
: > $out
b() {
  echo "== bench $1" >> $out
  timeout -k 10 300 env $1 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-configs 2>> $out | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value %.3f evals/s  seq %.2f ms  frac %.3f lZ %.9g' % (r['value'], r['sequential']['ms_per_eval'], r['roofline']['frac'], r['lZ_first']))
" >> $out
}
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
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=3"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=4 GPU_MAX_HW_QUEUES=8"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=3 GPU_MAX_HW_QUEUES=8"
b "BASE=1"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=3"
c4 "BASE=1"
c4 "GPX_BATCH_LOOKAHEAD=1"
c4 "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=2"
cat $out
