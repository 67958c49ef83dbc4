#!/usr/bin/env python3
"""Value-only evaluation, update + posterior (which use what the factorisation leaves behind)
and their times at a few sizes; run with GPX_PANEL_WHOLE=0 and =4096 and compare.
usage: whole_check.py N [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
for N in [int(a) for a in sys.argv[1:]]:
    D = 8
    X, y, Xs = recipes.synthetic(N, D, n_test=64)
    dev.set_data(X, y)
    th = recipes.theta_eval(D, 2)
    kk = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
    spec = kk._kspec()
    lZ = dev.exact_eval(spec, th[0], th[-1], False)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); dev.exact_eval(spec, th[0], th[-1], False); ts.append(time.perf_counter() - t0)
    tu = []
    for _ in range(12):
        t0 = time.perf_counter(); dev.exact_update(spec, th[0], th[-1]); tu.append(time.perf_counter() - t0)
    mu, s2 = dev.exact_posterior(Xs)
    pg = dev.exact_posterior_grad(Xs[:5])
    lg, dl = dev.exact_eval(spec, th[0], th[-1], True)
    print('%-20s N=%5d lZ %s | value %.3f ms update %.3f ms | mu %s s2 %s dmu %s | grad lZ %s' %
          (os.environ.get('GPX_PANEL_WHOLE', '-'), N, float(lZ).hex(), 1e3 * np.median(ts), 1e3 * np.median(tu),
           float(mu.sum()).hex(), float(s2.sum()).hex(), float(np.abs(pg[2]).sum()).hex(), float(lg).hex()), flush=True)
