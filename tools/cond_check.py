#!/usr/bin/env python3
"""Accuracy against the oracle as K gets ill-conditioned (small noise, smooth
kernel): lZ, dlZ and posterior errors. usage: cond_check.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from oracle import gp_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = 2
X, y, Xs = recipes.synthetic(N, D, n_test=50)
for sn in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
    gp = pygp_amd.BasicGP(sn, 1.0, [0.5, 0.7])
    gp.add_data(X, y)
    lZ, dlZ = gp.loglikelihood(True)
    mu, s2 = gp.posterior(Xs)
    spec = orc.se_spec(1.0, np.array([0.5, 0.7]))
    th = gp.get_hyper()
    K = orc.kernel_get(spec, X) + sn ** 2 * np.eye(N)
    cond = np.linalg.cond(K)
    R, a = orc.exact_update(spec, th[0], th[-1], X, y)
    wl, wd = orc.exact_loglik(spec, th[0], X, R, a, True)
    wm, ws = orc.exact_posterior(spec, th[-1], X, R, a, Xs)
    print('sn=%.0e cond(K)=%.1e  lZ rel err %.1e  dlZ rel err %.1e  mu err %.1e  s2 err %.1e'
          % (sn, cond, abs(lZ - wl) / abs(wl), np.max(np.abs(dlZ - wd)) / np.max(np.abs(wd)),
             np.max(np.abs(mu - wm)), np.max(np.abs(s2 - ws))), flush=True)
