#!/bin/bash
# PMC passes over the HBM-bound kernels (tools/run_hbm.py), one counter group per run
# as the gfx950 guide prescribes. usage (on the GPU box): tools/pmc_hbm.sh <tag>
set -o pipefail
tag=${1:-r02_hbm}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/run_hbm.py 3 > $out/trace.log 2>&1 || exit 1
i=0
while read -r group; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d $out/pmc$i -- python3 tools/run_hbm.py 1 > $out/pmc$i.log 2>&1 || exit 1
  echo "pass $i done: $group"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_SALU
SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
FETCH_SIZE
WRITE_SIZE
GROUPS
for d in $out/pmc*/; do
  f=$(ls $d/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py $f 1000000 | grep -A12 -E "kbuild|trace_grad" > $out/$(basename $d)_summary.txt
done
python3 tools/trace_summary.py $(ls $out/trace/*/*kernel_trace.csv | head -1) 1 > $out/trace_summary.txt
