#!/usr/bin/env python3
"""Posterior edge shapes: many test points on tiny models, one test point on large ones,
test points equal to training points, with input gradients; single models and batches, against
the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
from oracle import gp_oracle as orc
D = 2
k = pygp_amd.kernels.Matern(0.9, [0.7, 1.2], d=3)
spec0 = orc.matern_spec(0.9, [0.7, 1.2], d=3)
dev = _lib.Handle(0)
worst = 0.0
for N, M, B in [(5, 20000, 3), (1, 9000, 2), (130, 8193, 2), (3000, 1, 4), (700, 700, 5), (9000, 3, 2)]:
    X, y, Xs = recipes.synthetic(N, D, n_test=M, seed=N)
    Xs[: min(N, M)] = X[: min(N, M)]                 # coincident with training points
    k0, _ = k, None
    base = np.r_[np.log(0.15), k.get_hyper(), 0.1]
    th = base + 0.05 * np.random.RandomState(N).randn(B, base.size)
    dev.set_data(X, y)
    out = dev.posterior_batch(k._kspec(), th, Xs, grad=True)
    for b in (0, B - 1):
        sb = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[b][1:-1])
        R, a = orc.exact_update(sb, th[b][0], th[b][-1], X, y)
        want = orc.exact_posterior_grad(sb, th[b][-1], X, R, a, Xs)
        kb = k.copy(th[b][1:-1])
        dev.exact_update(kb._kspec(), th[b][0], th[b][-1])
        one = dev.exact_posterior_grad(Xs)
        for g, o, w in zip(out, one, want):
            e = max(np.max(np.abs(g[b] - w)), np.max(np.abs(o - w)))
            worst = max(worst, e)
            assert e <= 1e-6, (N, M, B, b, e)
    print('N=%d M=%d B=%d ok' % (N, M, B), flush=True)
print('worst abs error %.1e; posterior edges ok' % worst)
