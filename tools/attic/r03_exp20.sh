#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp20.log
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
b "BASE=1"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=2"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=3"
b "GPX_BATCH_LOOKAHEAD=1 GPX_BATCH_INFLIGHT=2 GPU_MAX_HW_QUEUES=8"
b "BASE=1"
cat $out
