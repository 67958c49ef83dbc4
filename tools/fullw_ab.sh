# A/B of the route "all of R^-1 inside the whole-matrix launch / sweep" (GPX_GRAD_FULL_W = largest
# padded order, 0 = never; GPX_PANEL_I128 = 0: bulk chunks as four 64x64 tasks, 1: one 128x128,
# default: 128x128 above 16 tiles): single evaluations with gradients (ms), groups of 8 in one
# launch and lock-step sweeps of 64 (evals/s)
set -e
for fw in 0 4096 0 4096; do
  echo "== GPX_GRAD_FULL_W=$fw"
  export GPX_GRAD_FULL_W=$fw
  for N in 1280 1536 2048 2560 3072 3584 4096; do timeout -k 10 100 python tools/route_time.py $N 12 | grep fused; done
  timeout -k 10 300 python tools/batch_small.py --b 8 --sizes 1536,2048,3072,4096 --reps 9 2>&1 | grep -o "\"n\": [0-9]*\|\"with_grad_evals_per_s\": [0-9.]*" | tr "\n" " "; echo
  timeout -k 10 300 python tools/batch_small.py --b 64 --sizes 1536,2048,3072,4096 --reps 5 2>&1 | grep -o "\"n\": [0-9]*\|\"with_grad_evals_per_s\": [0-9.]*" | tr "\n" " "; echo
done
