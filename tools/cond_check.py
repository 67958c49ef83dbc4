#!/usr/bin/env python3
"""Accuracy against the oracle as K gets ill-conditioned (small noise, smooth
kernel): lZ, dlZ and posterior errors of the model route (add_data -> update,
loglikelihood(True), posterior) and of the fused evaluation. usage: cond_check.py [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
from oracle import gp_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
D = 2
X, y, Xs = recipes.synthetic(N, D, n_test=50)
npad = (N + 127) // 128 * 128
print('N = %d (padded %d). Paths: update = %s; loglikelihood(True) after it = trtri (R^-1 '
      'completed from the inverses of the 1024-blocks) + lauum; fused = right-looking sweep '
      'with R^-1 and K^-1 built inside it (explicit-inverse row panels).'
      % (N, npad, 'ONE panel launch over the whole matrix, row panels by substitution'
         if 1024 < npad <= 4096 else
         ('one panel launch (a single block)' if npad <= 1024 else
          'right-looking sweep over %d 1024-blocks, explicit-inverse row panels '
          'R[k,k+1:] = W_kk^T A[k,k+1:]' % (npad // 1024))))
sns = (1e-1, 1e-2, 1e-3, 1e-4, 1e-5) if N <= 4096 else (1e-2, 1e-3, 1e-4)
dev = _lib.Handle(0)
dev.set_data(X, y)
for sn in sns:
    gp = pygp_amd.BasicGP(sn, 1.0, [0.5, 0.7])
    gp.add_data(X, y)
    lZ, dlZ = gp.loglikelihood(True)
    mu, s2 = gp.posterior(Xs)
    spec = orc.se_spec(1.0, np.array([0.5, 0.7]))
    th = gp.get_hyper()
    K = orc.kernel_get(spec, X) + sn ** 2 * np.eye(N)
    cond = np.linalg.cond(K) if N <= 4096 else np.abs(K).sum(0).max() / sn ** 2
    R, a = orc.exact_update(spec, th[0], th[-1], X, y)
    wl, wd = orc.exact_loglik(spec, th[0], X, R, a, True)
    wm, ws = orc.exact_posterior(spec, th[-1], X, R, a, Xs)
    kk = pygp_amd.kernels.SE(1.0, [0.5, 0.7])
    l2, d2 = dev.exact_eval(kk._kspec(), th[0], th[-1], True)
    print('sn=%.0e cond(K)%s%.1e  model: lZ rel err %.1e  dlZ rel err %.1e  mu err %.1e  s2 err %.1e'
          '  fused: lZ %.1e  dlZ %.1e'
          % (sn, '=' if N <= 4096 else '<=', cond, abs(lZ - wl) / abs(wl),
             np.max(np.abs(dlZ - wd)) / np.max(np.abs(wd)),
             np.max(np.abs(mu - wm)), np.max(np.abs(s2 - ws)),
             abs(l2 - wl) / abs(wl), np.max(np.abs(d2 - wd)) / np.max(np.abs(wd))), flush=True)
