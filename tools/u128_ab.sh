# far-tile update batches of the whole-matrix launch as ONE 128x128 task when the tile row is
# 3 + GPX_PANEL_U128 steps away (-1: never): one evaluation with gradients | value-only (ms)
set -e
for u in -1 4 6 8 12 -1 4 6 8 12; do
  echo "== GPX_PANEL_U128=$u"
  export GPX_PANEL_U128=$u
  for N in 2048 3072 4096; do timeout -k 10 100 python tools/seq_time.py $N 12; done
done
