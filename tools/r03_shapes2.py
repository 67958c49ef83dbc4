#!/usr/bin/env python3
"""round-3 probe: inverse-column product (NN, KLO_M) with k walked downwards (KREV)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
KLO_M, KREV = 2, 32
def live(M, N):
    return 2.0 * 128 * 128 * sum(M - m0 for m0 in range(0, M, 128)) * (N // 128)
tag = os.environ.get('TAG', '')
for M in (13312, 9216, 5120):
    for ta in (0, 1):
        for fl in (KLO_M, KLO_M | KREV):
            ms = dev.la_gemm_bench_mnk(M, 2048, M, ta, 0, fl, 0.0, reps=3)
            print('%-18s ta%d M=%6d fl=%2d %8.3f ms %6.2f TF' % (tag, ta, M, fl, ms, live(M, 2048) / ms * 1e-9), flush=True)
