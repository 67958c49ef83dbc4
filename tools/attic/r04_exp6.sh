#!/bin/bash
# C4 (64 thetas x N = 8192): one group of 64 / two of 32 one after the other / two in flight
for cfg in "32 2" "32 1" "64 1" "16 1" "16 2"; do
  set -- $cfg
  echo "== GPX_GROUP_MEMBERS=$1 GPX_GROUP_INFLIGHT=$2"
  GPX_GROUP_MEMBERS=$1 GPX_GROUP_INFLIGHT=$2 timeout -k 10 200 python tools/bench_configs.py c4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('value %.1f evals/s  grad %.1f evals/s' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s']))
"
done
