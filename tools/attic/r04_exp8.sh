#!/bin/bash
# task traces of the whole-matrix launch at N = 2048 (value-only): production, and with the
# three loops of the tile chain relaxed (timing only, wrong factor) -- throw-away switches
set -o pipefail
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04_exp8; mkdir -p $out
GPX_PANEL_DEBUG=2 python3 tools/attic/r04_skip_time.py 2048 > $out/base.log 2>&1 || exit 1
GPX_PANEL_DEBUG=2 GPX_EXP_NOWAIT=3 GPX_PANEL_LEAF_SKIP=64 python3 tools/attic/r04_skip_time.py 2048 > $out/relaxed.log 2>&1 || exit 1
for t in base relaxed; do
  python3 tools/panel_trace_summary.py $out/$t.log > $out/$t.txt 2>&1
  echo "== $t"; grep -n "panel trace" $out/$t.txt | tail -1
  awk '/panel trace/{n++} {if(n>=9) print}' $out/$t.txt | head -34
done
