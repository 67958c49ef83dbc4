#!/usr/bin/env python3
"""TFLOP/s of the product shapes the factorisation launches (rank-K updates, row
panels), in isolation. usage: gemm_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
UPPER, KLO_M, KHI_M, KLO_N, KHI_N = 1, 2, 4, 8, 16
def run(name, M, N, K, ta, tb, flags, beta, flop):
    ms = dev.la_gemm_bench_mnk(M, N, K, ta, tb, flags, beta, reps=3)
    print('%-44s M=%6d N=%6d K=%5d  %8.3f ms  %6.2f TFLOP/s' % (name, M, N, K, ms, flop / ms * 1e-9), flush=True)
for K in (512, 1024, 2048, 4096, 8192):
    n = 14336
    run('SYRK upper, beta=1 (ta=1,tb=0)', n, n, K, 1, 0, UPPER, 1.0, float(n) * n * K)
for K in (1024, 2048):
    n = 8192
    run('SYRK upper, beta=1 (ta=1,tb=0)', n, n, K, 1, 0, UPPER, 1.0, float(n) * n * K)
for K in (1024, 2048):
    run('full update, beta=1 (ta=1,tb=0)', 8192, 8192, K, 1, 0, 0, 1.0, 2.0 * 8192 * 8192 * K)
    run('full product, beta=0 (ta=1,tb=0)', 8192, 8192, K, 1, 0, 0, 0.0, 2.0 * 8192 * 8192 * K)
run('row panel W^T B (KHI_M)', 1024, 15360, 1024, 1, 0, KHI_M, 0.0, 1024.0 * 15360 * 1024)
run('row update (ta=1,tb=0) beta=1', 1024, 14336, 1024, 1, 0, 0, 1.0, 2.0 * 1024 * 14336 * 1024)
run('T = R12 W22 (KHI_N) n=4096', 4096, 4096, 4096, 0, 0, KHI_N, 0.0, 4096.0 ** 3)
run('W12 = -W11 T (KLO_M) n=4096', 4096, 4096, 4096, 0, 0, KLO_M, 0.0, 4096.0 ** 3)
run('T = R12 W22 (KHI_N) n=1024', 1024, 1024, 1024, 0, 0, KHI_N, 0.0, 1024.0 ** 3)
run('lauum n=16384', 16384, 16384, 16384, 0, 1, UPPER | KLO_M | KLO_N, 0.0, 16384.0 ** 3 / 3)
