#!/usr/bin/env python3
"""Stability / leak check: repeated handle creation, batches at several sizes,
posterior calls; device memory in use must return to its starting level."""
import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
hip = ctypes.CDLL('libamdhip64.so')
def used():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2 ** 20
D = 8
k = pygp_amd.kernels.SE(1.0, np.ones(D))
dev0 = _lib.Handle(0)          # keeps the runtime's own pools alive
base = used()
print('in use at start: %.0f MiB' % base, flush=True)
t0 = time.time()
for rep in range(12):
    dev = _lib.Handle(0)
    for N in (700, 3000, 8192) + ((12288,) if rep % 4 == 0 else ()):   # 12288: members with look-ahead
        X, y, Xs = recipes.synthetic(N, D, n_test=100)
        dev.set_data(X, y)
        th = np.array([recipes.theta_sweep(D, b + rep) for b in range(9)])
        lZ, dlZ = dev.loglik_batch(k._kspec(), th, grad=True)
        assert np.all(np.isfinite(lZ)) and np.all(np.isfinite(dlZ))
        mu, s2 = dev.posterior_batch(k._kspec(), th[:4], Xs)
        assert np.all(np.isfinite(mu)) and np.all(s2 > -1e-9)
        kk = k.copy(th[0][1:-1])
        dev.exact_update(kk._kspec(), th[0][0], th[0][-1])
        for _ in range(5):
            dev.exact_posterior_grad(Xs[:7])
    dev.close()
    del dev
    print('rep %2d: in use %.0f MiB (+%.0f), %.1f s' % (rep, used(), used() - base, time.time() - t0), flush=True)
print('leak: %.0f MiB' % (used() - base))
