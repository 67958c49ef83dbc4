#!/bin/bash
# C4 (64 thetas x N = 8192) against the group size / sweep switch
for cfg in "8 16" "16 16" "32 16" "16 99" "4 16"; do
  set -- $cfg
  echo "== GPX_GROUP_MEMBERS=$1 GPX_SWEEP_MIN_MEMBERS=$2"
  GPX_GROUP_MEMBERS=$1 GPX_SWEEP_MIN_MEMBERS=$2 timeout -k 10 200 python tools/bench_configs.py c4 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('value %.1f evals/s  grad %.1f evals/s  one %.2f ms  lZ0 %.10f' % (r['value_only_evals_per_s'], r['with_grad_evals_per_s'], r['one_eval_with_grad_ms'], r['lZ0']))
"
done
