#!/usr/bin/env python3
"""Groups with more than 16 input dimensions (D = 17, 24, 32: the widest instances of the build,
trace and posterior kernels) and Matern members: against single evaluations (bits) and the
oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
from oracle import gp_oracle as orc
dev = _lib.Handle(0)
for D in (17, 24, 32):
    for N, B in ((90, 4), (300, 20), (1500, 5)):
        ell = np.linspace(1.5, 3.0, D)
        for fam in ('se', 'matern'):
            k = pygp_amd.kernels.SE(1.0, ell) if fam == 'se' else pygp_amd.kernels.Matern(1.0, ell, d=5)
            spec0 = orc.se_spec(1.0, ell) if fam == 'se' else orc.matern_spec(1.0, ell, d=5)
            X, y, Xs = recipes.synthetic(N, D, n_test=6, seed=D + N)
            base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
            th = base + 0.05 * np.random.RandomState(D).randn(B, base.size)
            dev.set_data(X, y)
            lZ, dlZ = dev.loglik_batch(k._kspec(), th, grad=True)
            lZv = dev.loglik_batch(k._kspec(), th, grad=False)
            out = dev.posterior_batch(k._kspec(), th, Xs, grad=True)
            for b in (0, B - 1):
                kb = k.copy(th[b][1:-1])
                l1, d1 = dev.exact_eval(kb._kspec(), th[b][0], th[b][-1], True)
                assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), (D, N, fam, b)
                assert dev.exact_eval(kb._kspec(), th[b][0], th[b][-1], False) == lZv[b]
                sb = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[b][1:-1])
                R, a = orc.exact_update(sb, th[b][0], th[b][-1], X, y)
                want_lZ, want_dlZ = orc.exact_loglik(sb, th[b][0], X, R, a, True)
                assert abs(lZ[b] - want_lZ) <= 1e-8 * abs(want_lZ)
                assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * max(1, np.max(np.abs(want_dlZ)))
                want = orc.exact_posterior_grad(sb, th[b][-1], X, R, a, Xs)
                for g, w in zip(out, want):
                    assert np.max(np.abs(g[b] - w)) <= 1e-6, (D, N, fam)
    print('D=%d ok' % D, flush=True)
print('wide groups ok')
