#!/usr/bin/env python3
"""gpx_posterior_batch throughput: B models x N points at M test points (run twice: as is,
and with GPX_GROUP_MAX_NP=0 for one context and stream per member)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
dev = _lib.Handle(0)
for N, B, M in ((512, 256, 64), (1024, 256, 64), (1024, 256, 1024), (2048, 128, 256), (8192, 32, 512)):
    D = 8
    X, y, Xs = recipes.synthetic(N, D, n_test=M)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    dev.set_data(X, y)
    for grad in (False, True):
        dev.posterior_batch(k._kspec(), thetas, Xs, grad=grad)
        t0 = time.perf_counter()
        dev.posterior_batch(k._kspec(), thetas, Xs, grad=grad)
        t = time.perf_counter() - t0
        print('N=%d B=%d M=%d grad=%d: %.1f ms  %.0f models/s' % (N, B, M, grad, t * 1e3, B / t), flush=True)
