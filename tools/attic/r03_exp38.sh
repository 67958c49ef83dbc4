#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp38.log
: > $out
for args in "c4" "c2 c4" "c3 c4" "c4 c4"; do
echo "== $args" >> $out
GPX_TWIN_LOG=1 timeout -k 10 300 python3 tools/bench_configs.py $args 2>>$out | python3 -c "
import sys, json
for l in sys.stdin:
    try: c = json.loads(l)
    except Exception: continue
    if c['config'].startswith('C4'): print(round(c['value_only_evals_per_s'],1), round(c['with_grad_evals_per_s'],1), round(c['one_eval_with_grad_ms'],2))
" >> $out
done
cat $out
