#!/bin/bash
# three thetas per step at N = 16384: contexts on a plain pool (default), on round 3's masked
# pool, and as one group of three in lock-step
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_exp11; mkdir -p $out
run() {
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --per-gpu 3 --no-configs --no-cpu-baseline > $out/$1.json 2> $out/$1.err || exit 1
  python3 - $out/$1.json $1 <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print(sys.argv[2], 'evals/s %.3f'%b['value'], 'ms/step %.1f'%b['ms_per_step'], b['config'].get('batch_arrangement'))
P
}
run plain
GPX_TWIN_MASKED=1 run masked
GPX_GROUP_MIN_BIG=3 run group3
GPX_GROUP_MIN_BIG=2 run group_b3_min2
