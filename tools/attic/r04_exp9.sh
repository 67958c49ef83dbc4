#!/bin/bash
# members per group step at N = 16384: 6 (the bench default), 8, 12, 16 on the same box
set -o pipefail
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_exp9; mkdir -p $out
for cfg in "6 0" "8 0" "12 12" "16 16"; do
  set -- $cfg
  env=""; [ "$2" != "0" ] && export GPX_GROUP_MEMBERS=$2 || unset GPX_GROUP_MEMBERS
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --per-gpu $1 --no-configs --no-cpu-baseline > $out/per$1.json 2> $out/per$1.err || exit 1
  python3 - $out/per$1.json <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print('per', b['config']['thetas_per_gpu_per_step'], 'evals/s %.3f'%b['value'], 'ms/step %.1f'%b['ms_per_step'], 'frac %.4f'%b['roofline']['frac'], b['config'].get('batch_arrangement'))
P
done
