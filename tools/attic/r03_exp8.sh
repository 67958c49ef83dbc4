#!/bin/bash
# round-3: block lists for value-only sweeps at N=16384, batch depth for C4
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp8.log
: > $out
run() { TAG="$1" env $1 python3 tools/seq_time.py $2 5 >> $out 2>&1; }
run "BASE=1" 16384
run "GPX_BLOCKS=1024,2048,2048,2048,2048,2048,1024" 16384
run "GPX_BLOCKS=1024,2048,2048,2048,2048,1024" 16384
run "GPX_BLOCKS=1024,2048,3072,3072,2048,1024" 16384
run "GPX_BLOCKS=1024,1024,2048,2048,2048,2048,2048,1024" 16384
run "GPX_BLOCKS=1024,3072,3072,3072,2048,1024" 16384
for d in 3 4 6; do
  echo "== GPX_BATCH_INFLIGHT=$d" >> $out
  GPX_BATCH_INFLIGHT=$d python3 tools/bench_configs.py c4 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value-only %.1f evals/s  grad %.1f evals/s  one %.2f ms' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['one_eval_with_grad_ms']))
" >> $out
done
cat $out
