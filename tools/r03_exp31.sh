#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp31.log
: > $out
timeout -k 10 600 python -m pytest tests/test_gpu_gp.py -m gpu -x -q -k "large_batch_members or outrun" 2>&1 | tail -5 >> $out
echo "== without the wait" >> $out
GPX_TEST_NOFIX=1 timeout -k 10 600 python -m pytest tests/test_gpu_gp.py -m gpu -x -q -k "outrun" 2>&1 | tail -8 >> $out
cat $out
