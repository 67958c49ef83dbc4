#!/bin/bash
# block lists at N = 16384: default (1024, then 2048), and progressive lists
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_exp15; mkdir -p $out
run() {
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-configs --no-cpu-baseline > $out/$1.json 2> $out/$1.err || exit 1
  python3 - $out/$1.json "$1" <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print('%-28s evals/s %.3f  ms/step %.1f  frac %.4f  sequential %.2f ms' % (sys.argv[2], b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['sequential']['dense_ms_per_eval']))
P
}
run default
GPX_BLOCKS=1024,2048,3072 run 1024,2048,3072
GPX_BLOCKS=1024,1024,2048,3072 run 1024,1024,2048,3072
GPX_BLOCKS=1024,2048,2048,3072 run 1024,2048,2048,3072
GPX_BLOCKS=2048 run 2048
run default_again
