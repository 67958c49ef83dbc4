#!/bin/bash
# staggered batches in whole-matrix launches: value-only evaluations one at a time
for cfg in "0 8" "1 8" "1 4" "1 6" "1 12" "1 16"; do
  set -- $cfg
  for n in 2048 3072 4096; do
    TAG="stagger=$1 kbatch=$2" GPX_PANEL_STAGGER=$1 GPX_PANEL_KBATCH=$2 timeout -k 10 120 python tools/seq_time.py $n 8
  done
done
