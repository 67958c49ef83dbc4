set -e
for fw in 0 4096 0 4096; do   # (max padded order of the route; the r05 log used 0 / 1 = off / everywhere)
  echo "== GPX_GRAD_FULL_W=$fw"
  GPX_GRAD_FULL_W=$fw timeout -k 10 300 python tools/batch_small.py --b 256 --sizes 1536,2048 --reps 7 2>&1 | grep -o "\"n\": [0-9]*\|\"with_grad_evals_per_s\": [0-9.]*" | tr "\n" " "; echo
  GPX_GRAD_FULL_W=$fw timeout -k 10 300 python tools/batch_small.py --b 8 --sizes 1536,2048,3072,4096 --reps 9 2>&1 | grep -o "\"n\": [0-9]*\|\"with_grad_evals_per_s\": [0-9.]*" | tr "\n" " "; echo
done
