#!/usr/bin/env python3
"""round-3 probe: the inverse-column product W[:k,k] = -W[:k,:k] T (NN, KLO_M, tall and
narrow, k-ranges up to 13312) against the same shape with full k, other layouts and
equal-k variants, in isolation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
UPPER, KLO_M, KHI_M, KLO_N, KHI_N = 1, 2, 4, 8, 16
def live(M, N, K, flags, tile=128):
    tot = 0
    for m0 in range(0, M, tile):
        for n0 in range(0, N, tile):
            if flags & UPPER and n0 + tile <= m0: continue
            klo, khi = 0, K
            if flags & KLO_M: klo = max(klo, m0)
            if flags & KHI_M: khi = min(khi, m0 + tile)
            if flags & KLO_N: klo = max(klo, n0)
            if flags & KHI_N: khi = min(khi, n0 + tile)
            tot += max(0, khi - klo)
    return 2.0 * tile * tile * tot
def run(name, M, N, K, ta, tb, flags, beta=0.0):
    ms = dev.la_gemm_bench_mnk(M, N, K, ta, tb, flags, beta, reps=3)
    print('%-34s ta%d tb%d M=%6d N=%6d K=%6d fl=%2d  %8.3f ms  %6.2f TFLOP/s' %
          (name, ta, tb, M, N, K, flags, ms, live(M, N, K, flags) / ms * 1e-9), flush=True)
for M in (13312, 9216, 5120):
    run('inverse column (KLO_M)', M, 2048, M, 0, 0, KLO_M)
    run('same, full k', M, 2048, M, 0, 0, 0)
    run('same, K=2048', M, 2048, 2048, 0, 0, 0)
    run('TN full k', M, 2048, M, 1, 0, 0)
    run('NT full k', M, 2048, M, 0, 1, 0)
    run('TN KLO_M', M, 2048, M, 1, 0, KLO_M)
run('wide: N=13312 M=2048 NN KLO_N', 2048, 13312, 13312, 0, 0, KLO_N)
run('square NN KLO_M n=8192', 8192, 8192, 8192, 0, 0, KLO_M)
run('square NN full n=8192', 8192, 8192, 8192, 0, 0, 0)
