#!/usr/bin/env python3
"""N = 32768 (np = 32768): B = 2, 3 in groups against single evaluations (bits) and time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
k = pygp_amd.kernels.SE(1.0, np.ones(8))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
X, y, _ = recipes.synthetic(N, 8)
dev.set_data(X, y)
for B in (2, 3):
    th = np.array([recipes.theta_eval(8, 100 + b) for b in range(B)])
    for grad in (True, False):
        t0 = time.perf_counter(); out = dev.loglik_batch(k._kspec(), th, grad=grad); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); out = dev.loglik_batch(k._kspec(), th, grad=grad); t2 = time.perf_counter() - t0
        kb = k.copy(th[B - 1][1:-1])
        one = dev.exact_eval(kb._kspec(), th[B - 1][0], th[B - 1][-1], grad)
        same = (one[0] == out[0][B - 1] and np.array_equal(one[1], out[1][B - 1])) if grad else one == out[B - 1]
        print('%s N=%d B=%d grad=%d: first %.2f s, then %.2f evals/s, last member==single %s %s' % (os.environ.get('TAG', ''), N, B, grad, t1, B / t2, same, dev.batch_plan(B, grad)), flush=True)
