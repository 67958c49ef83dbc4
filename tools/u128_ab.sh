set -e
for u in -1 4 6 8 12 -1 4 6 8 12; do
  echo "== GPX_PANEL_U128=$u"
  export GPX_PANEL_U128=$u
  for N in 2048 3072 4096; do timeout -k 10 100 python tools/seq_time.py $N 12; done
done
