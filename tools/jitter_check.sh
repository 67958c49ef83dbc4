#!/bin/bash
# tools/jitter_check.py plain and under four seeds; prints the lines that differ
cd "$(dirname "$0")/.."
export GPX_MULTI_FAKE=1
out=gpurun_out/jitter
mkdir -p $out
timeout -k 10 600 python3 tools/jitter_check.py $1 > $out/plain.txt 2> $out/plain.err || { echo "plain run failed"; tail -5 $out/plain.err; exit 1; }
cat $out/plain.txt
rc=0
for seed in 11 12:800 13:100 14:2000; do
  GPX_TEST_JITTER=$seed timeout -k 10 900 python3 tools/jitter_check.py $1 > $out/seed_${seed%%:*}.txt 2> $out/seed_${seed%%:*}.err || { echo "seed $seed: run failed"; tail -5 $out/seed_${seed%%:*}.err; rc=1; continue; }
  if diff $out/plain.txt $out/seed_${seed%%:*}.txt > $out/diff_${seed%%:*}.txt; then echo "seed $seed: identical"; else echo "seed $seed: DIFFERENT"; cat $out/diff_${seed%%:*}.txt; rc=1; fi
done
exit $rc
