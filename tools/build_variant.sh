#!/bin/bash
# developer aid: a variant of libgpx.so with extra -D flags for ONE translation unit
# usage: tools/build_variant.sh NAME unit.hip "-DFOO=1 ..."   ->  pygp_amd/libgpx.so.NAME
# (on the GPU box: cp pygp_amd/libgpx.so.NAME pygp_amd/libgpx.so before the run)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; flags=$3
obj=pygp_amd/csrc/_obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c pygp_amd/csrc/$unit -o $obj/${unit%.hip}.$name.o
objs=""
for u in gpx_api kmat gemm_f64 chol leaf panel vec multi group; do
  if [ "$u.hip" == "$unit" ]; then objs="$objs $obj/$u.$name.o"; else objs="$objs $obj/$u.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o pygp_amd/libgpx.so.$name $objs -ldl
echo built pygp_amd/libgpx.so.$name
