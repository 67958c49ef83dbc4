#!/usr/bin/env python3
"""How a lone evaluation's time depends on what the GPU did right before (clock ramp after
idle): warm_time.py N [D] -> ms per call of update / log-lik+grad / value-only for the
calls 1-5, 6-20, 21-100, 101-300 of a back-to-back series that starts on an idle GPU."""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
total = int(sys.argv[3]) if len(sys.argv) > 3 else 300
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0)
dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
th = recipes.theta_eval(D, 0)
spec = k.copy(th[1:-1])._kspec()
calls = {'update': lambda: dev.exact_update(spec, th[0], th[-1]),
         'loglik+grad': lambda: dev.exact_eval(spec, th[0], th[-1], True),
         'value-only': lambda: dev.exact_eval(spec, th[0], th[-1], False)}
bins = [(0, 5), (5, 20), (20, 100), (100, total)]
for name, f in calls.items():
    f()                                           # allocations, task lists
    time.sleep(1.0)                               # idle GPU
    ts = []
    for i in range(total):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print('N=%5d %-12s ' % (N, name) +
          '  '.join('calls %d-%d: med %.3f min %.3f' % (a + 1, b, np.median(ts[a:b]), ts[a:b].min())
                    for a, b in bins if b <= total and a < b), flush=True)
