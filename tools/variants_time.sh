#!/bin/bash
# time the library variants built by tools/build_variant.sh: variants_time.sh "v1 v2 ..." [sizes]
cp pygp_amd/libgpx.so pygp_amd/libgpx.so.base
for v in base $1; do
  cp pygp_amd/libgpx.so.$v pygp_amd/libgpx.so
  for n in ${2:-1024 2048 4096}; do
    TAG=$v python tools/seq_time.py $n 12
  done
done
cp pygp_amd/libgpx.so.base pygp_amd/libgpx.so
