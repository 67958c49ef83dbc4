# factor on the nominal costs of the chain's tasks in the simulation that orders the queue of
# the whole-matrix launch (Graph::chain_us): one evaluation with gradients | value-only (ms)
set -e
for f in 1.0 0.6 0.7 0.8 1.0 0.6 0.7 0.8; do
  echo "== GPX_PANEL_CHAIN_SCALE=$f"
  export GPX_PANEL_CHAIN_SCALE=$f
  for N in 1536 2048 3072 4096; do timeout -k 10 100 python tools/seq_time.py $N 12; done
done
