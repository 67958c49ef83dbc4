#!/usr/bin/env python3
"""Open, use and close 150 handles in one process (lone evaluations, a batch in groups, a
batch through three contexts every tenth handle): time per cycle and free device memory."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
hip = C.CDLL('libamdhip64.so')
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2**20
k = pygp_amd.kernels.SE(1.0, np.ones(3))
t0 = time.time()
for i in range(150):
    N = [60, 300, 1100, 2300][i % 4]
    X, y, _ = recipes.synthetic(N, 3, seed=i)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    th = np.array([recipes.theta_sweep(3, b) for b in range(5)])
    dev.exact_eval(k.copy(th[0][1:-1])._kspec(), th[0][0], th[0][-1], True)
    dev.loglik_batch(k._kspec(), th, grad=bool(i & 1))
    dev.close()
    if i % 25 == 0:
        print('handle %3d: %.1f s, free %.0f MB' % (i, time.time() - t0, free_mb()), flush=True)
print('done %.1f s, free %.0f MB' % (time.time() - t0, free_mb()))
