#!/bin/bash
cd "$(dirname "$0")/.."
GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 timeout -k 10 120 python3 tools/panel_dbg.py 1024 1024 > gpurun_out/r03_ptrace_9.log 2>&1
grep -E "^  (xs|sub)" gpurun_out/r03_ptrace_9.log | tail -16
python3 tools/panel_trace_summary.py gpurun_out/r03_ptrace_9.log | tail -18
