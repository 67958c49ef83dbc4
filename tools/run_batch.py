#!/usr/bin/env python3
"""One (or a few) gpx_loglik_batch calls with gradients: B thetas at size N on the resident
data -- the target of rocprofv3 trace / PMC runs over the member-batched groups.
usage: run_batch.py [N] [B] [calls] [D] [grad 0/1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes
import pygp_amd
from pygp_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 1
D = int(sys.argv[4]) if len(sys.argv) > 4 else 8
grad = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0)
dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
print('plan', dev.batch_plan(B, grad=grad))
for c in range(calls):
    thetas = np.array([recipes.theta_eval(D, c * B + i) for i in range(B)])
    out = dev.loglik_batch(k._kspec(), thetas, grad=grad)
print('lZ', (out[0] if grad else out)[0])
