#!/bin/bash
# Round-5 collection on the GPU box (through gpurun): raw rocprofv3 outputs ->
# gpurun_out/<tag>/ (tools/make_profile_summaries_r04.py (it takes the tag) turns them into profiles/).
# One counter group per run and never together with a trace, as the gfx950 guide
# prescribes; the program itself follows `--` (no env / bash -c hop).
set -o pipefail
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
python3 bench.py --steps 10 --warmup 2 > $out/bench_n1.json 2> $out/bench_n1.err || exit 1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
echo "trace (bench: groups of six + sequential) done"
export GPX_GEMM_LOG=$out/gemmlog_seq.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_seq -- python3 tools/run_eval.py 16384 6 > $out/trace_seq.log 2>&1 || exit 1
unset GPX_GEMM_LOG
echo "trace (sequential) done"
# counters over ONE group of six members in lock-step (the timed region's unit of work)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/run_batch.py 16384 6 1 > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 tools/run_batch.py 16384 6 1 > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 tools/run_batch.py 16384 6 1 > $out/pmc_mfma.log 2>&1 || exit 1
echo "pmc (one group of six) done"
for n in 512 1024 2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/small_$n -- python3 tools/batch_small.py --b 256 --sizes $n --reps 1 > $out/small_$n.log 2>&1 || exit 1
done
echo "traces (256 thetas at N = 512, 1024, 2048) done"
GPX_PANEL_DEBUG=2 GPX_LOOKAHEAD=0 python3 tools/panel_dbg.py 1024 1024 1024 > $out/panel_trace.log 2>&1 || exit 1
echo "panel task trace done"
# round 5: the whole-matrix launch traced task by task at N = 2048 and 4096, and the same
# launch with the round-2 graph (fused spine task, every update a product task) beside it
GPX_PANEL_DEBUG=2 python3 tools/run_value.py 2048 3 > $out/panel_whole_2048.log 2>&1 || exit 1
GPX_PANEL_DEBUG=2 python3 tools/run_value.py 4096 3 > $out/panel_whole_4096.log 2>&1 || exit 1
GPX_PANEL_SPLIT=0 GPX_PANEL_DEBUG=2 python3 tools/run_value.py 2048 3 > $out/panel_whole_2048_fused.log 2>&1 || exit 1
# ... and evaluations with gradients: all of R^-1 inside the launch (default up to np = 2048; forced at 4096)
GPX_PANEL_DEBUG=2 python3 tools/run_value.py 2048 3 grad > $out/panel_whole_2048_grad.log 2>&1 || exit 1
GPX_GRAD_FULL_W=4096 GPX_PANEL_DEBUG=2 python3 tools/run_value.py 4096 3 grad > $out/panel_whole_4096_grad_fullw.log 2>&1 || exit 1
echo "whole-matrix traces done"
for n in 512 1024 2048 4096 8192; do python3 tools/seq_time.py $n 12; done > $out/seq_time.txt 2>&1
for n in 512 1024 2048 4096 8192; do GPX_PANEL_SPLIT=0 GPX_GRAD_WHOLE=0 TAG=round2-graph python3 tools/seq_time.py $n 12; done >> $out/seq_time.txt 2>&1
python3 tools/stage_time.py > $out/stage_time.txt 2>&1
echo "single evaluations done"
tail -c 300 $out/bench_n1.json
