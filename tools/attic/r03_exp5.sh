#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp5.log
: > $out
TAG=rowgrp0 GPX_TILE_ROWGRP=0 python3 tools/r03_shapes2.py >> $out 2>&1
TAG=rowgrp1 GPX_TILE_ROWGRP=1 python3 tools/r03_shapes2.py >> $out 2>&1
run() { TAG="$1" env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "GPX_KREV_INVCOL=0" 16384
run "GPX_KREV_INVCOL=1" 16384
run "GPX_KREV_INVCOL=0" 16384
run "GPX_KREV_INVCOL=1" 16384
run "GPX_KREV_INVCOL=1 GPX_KREV=1" 16384
run "GPX_KREV_INVCOL=1" 8192
run "GPX_KREV_INVCOL=0" 8192
cat $out
