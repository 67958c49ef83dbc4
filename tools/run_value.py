#!/usr/bin/env python3
"""Value-only evaluations one at a time (the slice-sampler pattern): run_value.py N [reps] [grad]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
N = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
grad = len(sys.argv) > 3 and sys.argv[3] == 'grad'
D = 8
X, y, _ = recipes.synthetic(N, D)
dev = _lib.Handle(0); dev.set_data(X, y)
k = pygp_amd.kernels.SE(1.0, np.ones(D))
for i in range(reps):
    th = recipes.theta_eval(D, i)
    lZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], grad)
print('lZ', lZ)
