#!/bin/bash
# round-3: conditioning through the multi-block driver
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r03_cond
python3 tools/backward_err.py 4096 > gpurun_out/r03_cond/backward_err_n4096.txt 2>&1 || exit 1
python3 tools/backward_err.py 1024 > gpurun_out/r03_cond/backward_err_n1024.txt 2>&1 || exit 1
python3 tools/cond_check.py 4096 > gpurun_out/r03_cond/cond_check_n4096.txt 2>&1 || exit 1
cat gpurun_out/r03_cond/*.txt
