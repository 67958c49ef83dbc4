#!/bin/bash
# round-3: HIP hardware queues (GPU_MAX_HW_QUEUES, default 4) against batch depth
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp10.log
: > $out
tools/bin/probe_i8 >> $out 2>&1
c4() {
  echo "== $1" >> $out
  env $1 python3 tools/bench_configs.py c4 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value-only %.1f evals/s  grad %.1f evals/s  one %.2f ms' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['one_eval_with_grad_ms']))
" >> $out
}
c4 "GPX_BATCH_INFLIGHT=3"
c4 "GPU_MAX_HW_QUEUES=8 GPX_BATCH_INFLIGHT=3"
c4 "GPU_MAX_HW_QUEUES=8 GPX_BATCH_INFLIGHT=4"
c4 "GPU_MAX_HW_QUEUES=8 GPX_BATCH_INFLIGHT=6"
c4 "GPU_MAX_HW_QUEUES=2 GPX_BATCH_INFLIGHT=3"
b() {
  echo "== bench $1" >> $out
  env $1 python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-configs 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value %.3f evals/s  seq %.2f ms  frac %.3f' % (r['value'], r['sequential']['ms_per_eval'], r['roofline']['frac']))
" >> $out
}
b "GPX_BATCH_INFLIGHT=3"
b "GPU_MAX_HW_QUEUES=8 GPX_BATCH_INFLIGHT=3"
b "GPU_MAX_HW_QUEUES=8 GPX_BATCH_INFLIGHT=4"
b "GPX_BATCH_INFLIGHT=2"
cat $out
