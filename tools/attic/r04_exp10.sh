#!/bin/bash
# bounded diagnostic: the tiny-datasets test, eight times through the groups (default) and
# eight times through the per-member contexts (GPX_GROUP_MAX_NP=0: three contexts on pool
# streams, a pool made and destroyed per case), each with a 30-s stack dump and a 90-s limit;
# stops at the first run that does not pass
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_exp10; mkdir -p $out
for i in 1 2 3 4 5 6 7 8; do
  for legacy in 16384 0; do
    GPX_GROUP_MAX_NP=$legacy GPX_DESTROY_LOG=1 timeout -k 5 90 python3 -X faulthandler -m pytest tests/test_gpu_groups.py -m gpu -x -v -k tiny -o faulthandler_timeout=30 -s > $out/run${i}_$legacy.log 2>&1
    rc=$?
    echo "run $i max_np=$legacy rc=$rc $(grep -c twin_pool_release $out/run${i}_$legacy.log) pool-stream destroys; $(tail -1 $out/run${i}_$legacy.log)"
    if [ $rc -ne 0 ]; then tail -40 $out/run${i}_$legacy.log | cut -c1-200; exit 1; fi
  done
done
