#!/usr/bin/env python3
"""The HBM-bound kernels of the path on their BASELINE sizes, a few launches each
(the target of tools/pmc_hbm.sh): kbuild fp32 SE+Periodic and SE at N=32768 D=4
(config 5), kbuild fp64 SE (upper tiles, inside an evaluation) and trace_grad at
N=16384 D=8. usage: run_hbm.py [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes
import pygp_amd
from pygp_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = _lib.Handle(0)
N, D = 32768, 4
X = np.random.RandomState(0).rand(N, D)
dev.set_data(X, np.zeros(N))
hse = _lib.KSpecHolder(_lib.KIND_SE, False, D, np.r_[0.0, np.log(np.linspace(.5, 1.5, D))])
hper = _lib.KSpecHolder(_lib.KIND_PERIODIC, False, D, np.r_[0.0, 0.0, np.log(0.7)])
hsum = _lib.KSpecHolder(_lib.KIND_SUM, False, D, parts=[hse, hper])
print('c5 se+periodic f32 ms', dev.kernel_build_resident(hsum, np.float32, reps=reps))
print('c5 se f32 ms', dev.kernel_build_resident(hse, np.float32, reps=reps))
print('c5 se f64 ms', dev.kernel_build_resident(hse, np.float64, reps=reps))
N, D = 16384, 8
X, y, _ = recipes.synthetic(N, D)
dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
for i in range(reps):
    th = recipes.theta_eval(D, i)
    lZ, dlZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
print('lZ', lZ)
# config 3's trace pass: Matern-5/2 ARD, D = 16
N, D = 16384, 16
X, y, _ = recipes.synthetic(N, D)
dev.set_data(X, y)
k = pygp_amd.kernels.Matern(1.0, np.ones(D), d=5)
for i in range(reps):
    th = recipes.theta_eval(D, i)
    lZ, dlZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
print('lZ matern D=16', lZ)
