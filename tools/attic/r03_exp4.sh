#!/bin/bash
# round-3 A/B: row-grouped XCD order of tall launches (GPX_TILE_ROWGRP)
cd "$(dirname "$0")/.."
out=gpurun_out/r03_exp4.log
: > $out
run() { TAG="$1" env $1 python3 tools/seq_time.py $2 6 >> $out 2>&1; }
run "GPX_TILE_ROWGRP=0" 16384
run "GPX_TILE_ROWGRP=1" 16384
run "GPX_TILE_ROWGRP=0" 16384
run "GPX_TILE_ROWGRP=1" 16384
run "GPX_TILE_ROWGRP=0" 8192
run "GPX_TILE_ROWGRP=1" 8192
cat $out
