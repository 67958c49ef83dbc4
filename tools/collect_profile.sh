#!/bin/bash
# Run on the GPU box (through gpurun): raw rocprofv3 outputs -> gpurun_out/<tag>/
# usage: tools/collect_profile.sh <tag> [steps]
set -o pipefail
tag=${1:-r01}; steps=${2:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python3 bench.py --steps 10 --warmup 2 > $out/bench_n1.json 2> $out/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_seq -- python3 tools/run_eval.py 16384 6 > $out/trace_seq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/run_eval.py 16384 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/run_eval.py 16384 1 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 tools/run_eval.py 16384 1 > $out/pmc_mfma.log 2>&1
cat $out/bench_n1.json
