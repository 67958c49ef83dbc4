#!/usr/bin/env python3
"""Developer probe: GEMM TFLOP/s, potrf(+inverse) times and per-stage times of
one log-lik+grad evaluation. Usage: python tools/quick_perf.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes
import pygp_amd
from pygp_amd import _lib

sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
dev = _lib.Handle(0)

for n in (2048, 4096, 8192):
    for ta, tb in ((1, 0), (0, 0), (0, 1)):
        ms = dev.la_gemm_bench(n, ta, tb, reps=3)
        print('GEMM n=%5d ta=%d tb=%d: %8.3f ms  %6.2f TFLOP/s' %
              (n, ta, tb, ms, 2.0 * n ** 3 / ms * 1e-9), flush=True)

for n in sizes:
    ms = dev.la_potrf_bench(n, False, reps=2)
    print('POTRF       n=%5d: %8.2f ms  %6.2f TFLOP/s (n^3/3)' %
          (n, ms, n ** 3 / 3.0 / ms * 1e-9), flush=True)
    ms = dev.la_potrf_bench(n, True, reps=2)
    print('POTRF+POTRI n=%5d: %8.2f ms  %6.2f TFLOP/s (n^3)' %
          (n, ms, float(n) ** 3 / ms * 1e-9), flush=True)

for n in sizes:
    D = 8
    X, y, _ = recipes.synthetic(n, D)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev.set_data(X, y)
    dev.enable_timing(True)
    for i in range(3):
        th = recipes.theta_eval(D, i)
        kk = k.copy(th[1:-1])
        t0 = time.time()
        lZ, dlZ = dev.exact_eval(kk._kspec(), th[0], th[-1], True)
        dt = time.time() - t0
    t = dev.timings()
    print('EVAL n=%5d: %.2f ms wall  lZ=%.10g' % (n, dt * 1e3, lZ))
    print('   ' + '  '.join('%s=%.2f' % (a, b) for a, b in t.items()), flush=True)
    t0 = time.time()
    lZ = dev.exact_eval(kk._kspec(), th[0], th[-1], False)
    print('EVAL value-only n=%5d: %.2f ms' % (n, (time.time() - t0) * 1e3), flush=True)
    dev.enable_timing(False)
