#!/bin/bash
# Run on the GPU box (through gpurun): raw rocprofv3 outputs -> gpurun_out/<tag>/
# usage: tools/collect_profile.sh <tag> [steps]
# One counter group per run and never together with a trace, as the gfx950 guide
# prescribes; the program itself follows `--` (no env / bash -c hop).
set -o pipefail
tag=${1:-r02}; steps=${2:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python3 bench.py --steps 10 --warmup 2 > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-configs > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
echo "trace (batched) done"
export GPX_GEMM_LOG=$out/gemmlog_seq.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_seq -- python3 tools/run_eval.py 16384 6 > $out/trace_seq.log 2>&1 || exit 1
unset GPX_GEMM_LOG
echo "trace (sequential) done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/run_eval.py 16384 1 > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/run_eval.py 16384 1 > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 tools/run_eval.py 16384 1 > $out/pmc_mfma.log 2>&1 || exit 1
echo "pmc (evaluation) done"
tools/pmc_hbm.sh $tag/hbm || exit 1
echo "pmc (hbm-bound kernels) done"
GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 python3 tools/panel_dbg.py 1024 1024 1024 > $out/panel_trace.log 2>&1 || exit 1
GPX_PANEL_STREAM=0 GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 python3 tools/panel_dbg.py 1024 1024 > $out/panel_trace_r1graph.log 2>&1 || exit 1
echo "panel task trace done"
python3 tools/bench_configs.py --out $out/configs.json > $out/configs.log 2>&1 || exit 1
tail -c 400 $out/bench_n1.json
