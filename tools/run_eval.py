#!/usr/bin/env python3
"""Run a few log-lik+grad evaluations at size N (no torch, no CPU baseline):
the target of rocprofv3 PMC / trace runs. usage: run_eval.py [N] [evals] [D]"""
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
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D = int(sys.argv[3]) if len(sys.argv) > 3 else 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0)
dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
for i in range(evals):
    th = recipes.theta_eval(D, i)
    lZ, dlZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
print('lZ', lZ)
