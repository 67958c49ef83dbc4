#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp9.log
: > $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -x -q > gpurun_out/r03_gputests5.log 2>&1; tail -3 gpurun_out/r03_gputests5.log >> $out
for w in 4 8; do
  echo "== GPX_KBUILD_W=$w" >> $out
  GPX_KBUILD_W=$w python3 tools/bench_configs.py c5 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('se+per f32 %.3f ms  se f32 %.3f ms  se f64 %.3f ms' % (r['se+periodic_fp32_ms'], r['se_fp32_ms'], r['se_fp64_ms']))
" >> $out
  GPX_KBUILD_W=$w python3 tools/quick_perf.py 16384 2>&1 | grep "kernel_build" >> $out
done
cat $out
