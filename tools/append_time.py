#!/usr/bin/env python3
"""time of appending one observation in place vs refactorising: append_time.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16300
D = 8
X, y, _ = recipes.synthetic(N + 20, D)
gp = pygp_amd.BasicGP(0.1, 1.0, np.linspace(.5, 1.5, D))
t0 = time.perf_counter(); gp.add_data(X[:N], y[:N]); t_full = time.perf_counter() - t0
t0 = time.perf_counter(); gp.set_hyper(gp.get_hyper()); t_full2 = time.perf_counter() - t0
ts = []
for i in range(N, N + 10):
    t0 = time.perf_counter(); gp.add_data(X[i:i + 1], y[i:i + 1]); ts.append(time.perf_counter() - t0)
print('N=%d: first add_data %.1f ms, refactorisation %.1f ms, in-place append of one point %.2f ms (median of 10)'
      % (N, t_full * 1e3, t_full2 * 1e3, np.median(ts) * 1e3))
