#!/usr/bin/env python3
"""evals/s of gpx_loglik_batch at size N (B thetas) for the in-flight depth given by
GPX_BATCH_INFLIGHT. usage: inflight_exp.py N B"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
N, B = int(sys.argv[1]), int(sys.argv[2])
D = 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0); dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
th = np.array([recipes.theta_sweep(D, b) for b in range(B)])
dev.loglik_batch(k._kspec(), th[:8], grad=True)
out = []
for g in (False, True):
    t0 = time.perf_counter(); dev.loglik_batch(k._kspec(), th, grad=g); t = time.perf_counter() - t0
    out.append('%s %.1f evals/s' % ('grad' if g else 'value', B / t))
print('N=%d inflight=%s: %s' % (N, os.environ.get('GPX_BATCH_INFLIGHT', '3'), '  '.join(out)))
