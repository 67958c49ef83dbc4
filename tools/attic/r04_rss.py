#!/usr/bin/env python3
"""Host memory over 20 000 small evaluations and 2 000 batches on one handle (RSS in MiB)."""
import os, sys, time
import numpy as np, psutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
p = psutil.Process()
dev = _lib.Handle(0)
k = pygp_amd.kernels.SE(1.0, np.ones(3))
X, y, Xs = recipes.synthetic(300, 3, n_test=5)
dev.set_data(X, y)
th = np.array([recipes.theta_sweep(3, b) for b in range(40)])
def rss(): return p.memory_info().rss / 2**20
print('start %.1f MiB' % rss(), flush=True)
for rep in range(5):
    for i in range(4000):
        t = th[i % 40]
        dev.exact_eval(k.copy(t[1:-1])._kspec(), t[0], t[-1], bool(i & 1))
    for i in range(400):
        dev.loglik_batch(k._kspec(), th, grad=bool(i & 1))
        if i % 10 == 0: dev.posterior_batch(k._kspec(), th, Xs)
    print('pass %d: %.1f MiB' % (rep, rss()), flush=True)
