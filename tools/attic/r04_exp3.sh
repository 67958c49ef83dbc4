#!/bin/bash
# near queue of whole-matrix launches: value-only evaluations one at a time
for cfg in "0 24" "3 24" "3 16" "3 40" "2 24" "4 32"; do
  set -- $cfg
  for n in 2048 3072 4096; do
    TAG="near=$1 wg=$2" GPX_PANEL_NEAR=$1 GPX_PANEL_NEARWG=$2 timeout -k 10 120 python tools/seq_time.py $n 8
  done
done
