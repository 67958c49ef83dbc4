#!/bin/bash
# bulk queue of whole-matrix launches (long batched updates taken only by flexible workers,
# which look before they claim): value-only evaluations one at a time
for cfg in "0 96" "4 96" "4 160" "4 200" "4 230" "2 200" "8 200"; do
  set -- $cfg
  for n in 2048 3072 4096; do
    TAG="bulk=$1 wg=$2" GPX_PANEL_BULK=$1 GPX_PANEL_BULKWG=$2 timeout -k 10 120 python tools/seq_time.py $n 8
  done
done
