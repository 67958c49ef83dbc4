#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp16.log
: > $out
timeout -k 10 900 python -m pytest tests/test_gpu_la.py tests/test_gpu_gp.py -m gpu -x -q > gpurun_out/r03_gputests9.log 2>&1
tail -4 gpurun_out/r03_gputests9.log >> $out
run() { TAG="$1" timeout -k 10 120 env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
for n in 2048 4096 8192 16384; do
  run "GPX_CHAIN_SPLITK=0" $n
  run "GPX_CHAIN_SPLITK=1" $n
done
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
c4 "GPX_CHAIN_SPLITK=0"
c4 "GPX_CHAIN_SPLITK=1"
cat $out
