#!/usr/bin/env python3
"""One-line timing of sequential evaluations (developer A/B probe for environment
switches): seq_time.py N [evals] [D] -> median / min ms of log-lik+grad and of a
value-only evaluation, one at a time."""
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0)
dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
out = {}
for grad in (True, False):
    ts = []
    for i in range(evals + 2):
        th = recipes.theta_eval(D, i)
        spec = k.copy(th[1:-1])._kspec()
        t0 = time.perf_counter()
        r = dev.exact_eval(spec, th[0], th[-1], grad)
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = ts[2:]
    out[grad] = (np.median(ts), np.min(ts))
tag = os.environ.get('TAG', '')
print('%-28s N=%5d grad med %.2f min %.2f | value med %.2f min %.2f | lZ %.10g' %
      (tag, N, out[True][0], out[True][1], out[False][0], out[False][1],
       r if np.isscalar(r) else r[0]), flush=True)
