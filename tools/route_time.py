#!/usr/bin/env python3
"""log-lik + gradient at N: the fused evaluation (gpx_exact_eval, sweep with the inverse in it
above np = 1024) against update + loglikelihood(True) (whole-matrix launch, then R^-1 completed
and K^-1 formed by trtri / lauum): route_time.py N [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
N = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
D = 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0); dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
def fused(i):
    th = recipes.theta_eval(D, i); s = k.copy(th[1:-1])._kspec()
    return dev.exact_eval(s, th[0], th[-1], True)
def two(i):
    th = recipes.theta_eval(D, i); s = k.copy(th[1:-1])._kspec()
    dev.exact_update(s, th[0], th[-1])
    return dev.exact_loglik(s.c.nhyper, True)
for name, f in (('fused', fused), ('update+loglik', two)):
    f(0); f(1)
    ts = []
    for i in range(reps):
        t0 = time.perf_counter(); r = f(i); ts.append((time.perf_counter() - t0) * 1e3)
    print('N=%5d %-14s med %.3f min %.3f ms  lZ %.12g dlZ[1] %.10g' % (N, name, np.median(ts), min(ts), r[0], r[1][1]), flush=True)
