#!/usr/bin/env python3
"""Batched evaluations per second at size N (developer A/B probe): batch_time.py N [B]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
N = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 9
D = 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0); dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
th = np.array([recipes.theta_eval(D, i) for i in range(B)])
out = []
for grad in (True, False):
    dev.loglik_batch(k._kspec(), th[:3], grad=grad)
    t0 = time.perf_counter(); r = dev.loglik_batch(k._kspec(), th, grad=grad); dt = time.perf_counter() - t0
    out.append(B / dt)
print('%-44s N=%5d B=%d grad %.2f evals/s | value-only %.2f evals/s | lZ0 %.10g' %
      (os.environ.get('TAG', ''), N, B, out[0], out[1], (r if not isinstance(r, tuple) else r[0])[0]), flush=True)
