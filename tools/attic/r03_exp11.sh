#!/bin/bash
# round-3: streams of the batch contexts chosen by the overlap probe vs the round-2 modes
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp11.log
: > $out
c4() {
  echo "== $1" >> $out
  env $1 python3 tools/bench_configs.py c4 2>> $out | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value-only %.1f evals/s  grad %.1f evals/s  one %.2f ms' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['one_eval_with_grad_ms']))
" >> $out
}
c4 "GPX_TWIN_STREAMS=pool"
c4 "GPX_TWIN_STREAMS=maskq"
c4 "GPX_TWIN_STREAMS=plain"
c4 "GPX_TWIN_STREAMS=pool"
c4 "GPX_TWIN_STREAMS=maskq"
b() {
  echo "== bench $1" >> $out
  env $1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-configs 2>> $out | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value %.3f evals/s  seq %.2f ms  frac %.3f' % (r['value'], r['sequential']['ms_per_eval'], r['roofline']['frac']))
" >> $out
}
b "GPX_TWIN_STREAMS=pool"
b "GPX_TWIN_STREAMS=maskq"
b "GPX_TWIN_STREAMS=plain"
cat $out
