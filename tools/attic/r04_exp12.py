#!/usr/bin/env python3
"""B = 2 and 3 thetas above np = 8192 (N = 9000, 12000, 16384; value-only and with
gradients): evals/s of the batch, median of 5 batches after a warm-up."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
k = pygp_amd.kernels.SE(1.0, np.ones(8))
for N in (9000, 12000, 16384):
    X, y, _ = recipes.synthetic(N, 8)
    dev.set_data(X, y)
    for B in (2, 3):
        th = np.array([recipes.theta_eval(8, 100 + b) for b in range(B)])
        for grad in (False, True):
            dev.loglik_batch(k._kspec(), th, grad=grad)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); dev.loglik_batch(k._kspec(), th, grad=grad); ts.append(time.perf_counter() - t0)
            print('%s N=%5d B=%d grad=%d: %.2f evals/s  %s' % (os.environ.get('TAG', ''), N, B, grad, B / np.median(ts), dev.batch_plan(B, grad)), flush=True)
