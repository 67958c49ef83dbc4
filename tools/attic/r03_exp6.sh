#!/bin/bash
# round-3: sequential kernel trace + launch log with and without the reversed k walk of
# the inverse-column product
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 1; do
  out=gpurun_out/r03_exp6_$v; mkdir -p $out
  export GPX_KREV_INVCOL=$v
  export GPX_GEMM_LOG=$out/gemmlog.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/run_eval.py 16384 4 > $out/trace.log 2>&1 || exit 1
  unset GPX_GEMM_LOG
  f=$(ls $out/trace/*/*kernel_trace.csv | head -1)
  python3 tools/gemm_trace_join.py $out/gemmlog.txt $f 300 > $out/launches.txt
  python3 tools/timeline.py $out/gemmlog.txt $f 2 1 > $out/timeline.txt
done
grep "fl= 2" gpurun_out/r03_exp6_0/launches.txt | tail -8
echo ---
grep "fl= 2\|fl=34" gpurun_out/r03_exp6_1/launches.txt | tail -8
tail -3 gpurun_out/r03_exp6_0/timeline.txt gpurun_out/r03_exp6_1/timeline.txt
