#!/bin/bash
# Run on the GPU box (through gpurun): the metric config's timed region (one group of six at
# N = 16384) with the plain longest-first tile order of batches (GPX_TILE_XCD=0, the default
# there) against the XCD-aware order (8x8 macro tiles per L2, GPX_TILE_XCD=1): evals/s, and
# the HBM traffic of one group step from two PMC passes each (FETCH_SIZE, WRITE_SIZE).
set -o pipefail
tag=${1:-xcd_ab}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for x in 0 1 0 1; do
  GPX_TILE_XCD=$x python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-configs > $out/bench_x$x.json 2> $out/bench_x$x.err || exit 1
  python3 - $out/bench_x$x.json $x <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('GPX_TILE_XCD=%s  %.3f evals/s  %.2f ms/step  dense frac %.4f  sequential %.2f ms' % (
    sys.argv[2], r['value'], r['ms_per_step'], r['roofline']['frac'], r['sequential']['ms_per_eval']))
PY
done
for x in 0 1; do
  export GPX_TILE_XCD=$x
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_x$x -- python3 tools/run_batch.py 16384 6 1 > $out/pmc_fetch_x$x.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_x$x -- python3 tools/run_batch.py 16384 6 1 > $out/pmc_write_x$x.log 2>&1 || exit 1
  unset GPX_TILE_XCD
  python3 - $out $x <<'PY'
import csv, glob, sys
out, x = sys.argv[1], sys.argv[2]
tot = {}
for name in ('fetch', 'write'):
    f = glob.glob('%s/pmc_%s_x%s/*/*counter_collection.csv' % (out, name, x))[0]
    s = 0.0
    for r in csv.DictReader(open(f)):
        s += float(r['Counter_Value'])
    tot[name] = s
# (KiB units; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> doubled)
b = (2 * tot['fetch'] + tot['write']) * 1024 / 6
print('GPX_TILE_XCD=%s  HBM traffic per evaluation %.1f GB (fetch %.1f GB, write %.1f GB)' % (
    x, b / 1e9, 2 * tot['fetch'] * 1024 / 6e9, tot['write'] * 1024 / 6e9))
PY
  rm -rf $out/pmc_fetch_x$x $out/pmc_write_x$x
done
