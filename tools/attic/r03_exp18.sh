#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp18.log
: > $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 8 >> $out 2>&1; }
for i in 1 2 3; do
  run "BASE=1" 16384
  run "GPX_GEMM_NOSPLIT=1" 16384
done
run "BASE=1" 8192
run "GPX_GEMM_NOSPLIT=1" 8192
b() {
  echo "== bench $1" >> $out
  env $1 python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-configs 2>> $out | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('value %.3f evals/s  seq %.2f ms  frac %.3f' % (r['value'], r['sequential']['ms_per_eval'], r['roofline']['frac']))
" >> $out
}
b "BASE=1"
b "GPX_GEMM_NOSPLIT=1"
b "BASE=1"
b "GPX_GEMM_NOSPLIT=1"
cat $out
