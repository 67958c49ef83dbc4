#!/bin/bash
# look-ahead inside the group of six at N = 16384
for cfg in "0 8" "1 8" "1 4" "1 16" "1 32"; do
  set -- $cfg
  echo "== GPX_GROUP_LOOKAHEAD=$1 GPX_PANEL_MWG_LA=$2"
  GPX_GROUP_LOOKAHEAD=$1 GPX_PANEL_MWG_LA=$2 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    print('value %.2f evals/s  ms/step %.1f  frac %.4f  lZ %.10f' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['lZ_first']))
"
done
