cp pygp_amd/libgpx.so pygp_amd/libgpx.so.base
for v in base head base head; do
  cp pygp_amd/libgpx.so.$v pygp_amd/libgpx.so
  python tools/batch_small.py --b 256 --sizes 512,1024,2048 --reps 9 2>&1 | grep -o "\"n\": [0-9]*\|\"value_only_evals_per_s\": [0-9.]*\|\"with_grad_evals_per_s\": [0-9.]*" | tr "\n" " " | sed "s/^/$v /"; echo
done
cp pygp_amd/libgpx.so.base pygp_amd/libgpx.so
