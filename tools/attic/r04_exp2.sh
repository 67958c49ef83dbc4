#!/bin/bash
# headline batch (6 thetas x N = 16384 per step) through the member-batched groups instead
# of three contexts with their own streams
for cfg in "0 0 2" "16384 6 1" "16384 3 2" "16384 2 3" "16384 6 2"; do
  set -- $cfg
  echo "== GPX_GROUP_MAX_NP=$1 GPX_GROUP_MEMBERS=$2 GPX_GROUP_INFLIGHT=$3"
  GPX_GROUP_MAX_NP=$1 GPX_GROUP_MEMBERS=$2 GPX_GROUP_INFLIGHT=$3 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('value %.2f evals/s  ms/step %.1f  seq %.2f ms  lZ %.10f' % (r['value'], r['ms_per_step'], r['sequential']['ms_per_eval'], r['lZ_first']))
"
done
