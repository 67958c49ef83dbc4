#!/usr/bin/env python3
"""B = 2 and 3 thetas at N = 20000 and 24000 (above np = 16384): contexts vs groups
(GPX_GROUP_MAX_NP=32768), with gradients and value-only; plus bit-equality with singles."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
k = pygp_amd.kernels.SE(1.0, np.ones(8))
for N in (20000, 24000):
    X, y, _ = recipes.synthetic(N, 8)
    dev.set_data(X, y)
    for B in (2, 3):
        th = np.array([recipes.theta_eval(8, 100 + b) for b in range(B)])
        for grad in (False, True):
            out = dev.loglik_batch(k._kspec(), th, grad=grad)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); dev.loglik_batch(k._kspec(), th, grad=grad); ts.append(time.perf_counter() - t0)
            kb = k.copy(th[0][1:-1])
            one = dev.exact_eval(kb._kspec(), th[0][0], th[0][-1], grad)
            same = (one[0] == out[0][0] and np.array_equal(one[1], out[1][0])) if grad else one == out[0]
            print('%s N=%5d B=%d grad=%d: %.2f evals/s  member0==single %s  %s' % (os.environ.get('TAG', ''), N, B, grad, B / np.median(ts), same, dev.batch_plan(B, grad)), flush=True)
