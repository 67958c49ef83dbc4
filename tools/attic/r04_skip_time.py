#!/usr/bin/env python3
"""Timing-only probe: value-only evaluations with errors ignored (used with
GPX_PANEL_LEAF_SKIP bits that make the factorisation wrong on purpose)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
for N in [int(a) for a in sys.argv[1:]] or [2048, 4096]:
    X, y, _ = recipes.synthetic(N, 8)
    dev.set_data(X, y)
    k = pygp_amd.kernels.SE(1.0, np.ones(8))
    ts, err = [], 0
    for i in range(10):
        th = recipes.theta_eval(8, i)
        spec = k.copy(th[1:-1])._kspec()
        t0 = time.perf_counter()
        try:
            dev.exact_eval(spec, th[0], th[-1], False)
        except Exception:
            err += 1
        ts.append((time.perf_counter() - t0) * 1e3)
    print('%s N=%d value med %.3f min %.3f ms (errors %d)' % (os.environ.get('TAG', ''), N, np.median(ts[2:]), np.min(ts[2:]), err), flush=True)
