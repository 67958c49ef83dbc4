#!/bin/bash
cd "$(dirname "$0")/.."
GPX_PANEL_DEBUG=2 timeout -k 10 300 python3 tools/run_eval.py 16384 2 > gpurun_out/r03_insitu.log 2>&1
python3 tools/panel_insitu.py gpurun_out/r03_insitu.log | tail -20
ls -la gpurun_out/r03_insitu.log
