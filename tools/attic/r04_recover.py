#!/usr/bin/env python3
"""Failures in the middle of large evaluations (not positive definite at N = 5000 / 9000, with
and without gradients, alone and inside batches) followed by good evaluations on the SAME
handle: the good ones must have the bits of a fresh handle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
D = 3
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
for N in (5000, 9000):
    X, y, Xs = recipes.synthetic(N, D, n_test=7)
    good = np.array([recipes.theta_sweep(D, b) for b in range(4)])
    bad = good.copy()
    bad[:, 0] = np.log(1e-9)                  # no noise
    bad[:, 2:2 + D] = np.log(50.0)            # nearly rank one
    fresh = _lib.Handle(0); fresh.set_data(X, y)
    want = [fresh.exact_eval(k.copy(t[1:-1])._kspec(), t[0], t[-1], True) for t in good]
    wantv = [fresh.exact_eval(k.copy(t[1:-1])._kspec(), t[0], t[-1], False) for t in good]
    wantb = fresh.loglik_batch(k._kspec(), good, grad=True)
    fresh.close()
    dev = _lib.Handle(0); dev.set_data(X, y)
    nfail = 0
    for i, t in enumerate(good):
        for grad in (True, False):
            try:
                dev.exact_eval(k.copy(bad[i][1:-1])._kspec(), bad[i][0], bad[i][-1], grad)
            except np.linalg.LinAlgError:
                nfail += 1
            got = dev.exact_eval(k.copy(t[1:-1])._kspec(), t[0], t[-1], grad)
            if grad:
                assert got[0] == want[i][0] and np.array_equal(got[1], want[i][1]), (N, i, grad)
            else:
                assert got == wantv[i], (N, i, grad)
    mixed = np.r_[good[:2], bad[:2], good[2:]]
    lZ, dlZ = dev.loglik_batch(k._kspec(), mixed, grad=True)
    ok = [0, 1, 4, 5]
    assert np.array_equal(lZ[ok], wantb[0]) and np.array_equal(dlZ[ok], wantb[1]), N
    assert np.all(np.isinf(lZ[[2, 3]])) and np.all(np.isnan(dlZ[[2, 3]]))
    got = dev.exact_eval(k.copy(good[0][1:-1])._kspec(), good[0][0], good[0][-1], True)
    assert got[0] == want[0][0] and np.array_equal(got[1], want[0][1])
    dev.close()
    print('N=%d: %d failures raised, every good evaluation after them bit-equal to a fresh handle' % (N, nfail), flush=True)
print('recover ok')
