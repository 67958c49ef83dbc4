#!/bin/bash
# build libgpx.so here (hipcc cross-compiles), then run a command on the GPU box
# usage: tools/gpu.sh <timeout_s> '<command>'
cd "$(dirname "$0")/.."
python3 -m pygp_amd.build || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
