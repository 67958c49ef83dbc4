#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp31.log
: > $out
GPX_TEST_NOFIX=1 timeout -k 10 900 python -m pytest tests/test_gpu_gp.py -m gpu -x -q -k "launch_timing" 2>&1 | tail -6 >> $out
cat $out
