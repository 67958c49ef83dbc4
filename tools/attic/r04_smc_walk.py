#!/usr/bin/env python3
"""Random SMC-style walks over HyperEnsemble (the loop of /root/reference/pygp/meta/smc.py:86-126
with the per-particle loops as batched device calls): data arrive in chunks of random length
from EMPTY, particles are reweighted, resampled when the effective sample size drops, moved;
after every step weights and the mixture posterior against the same bookkeeping done with the
oracle per particle."""
import os, sys, time
import numpy as np
from scipy.special import logsumexp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd.meta import HyperEnsemble
from pygp_amd.likelihoods import Gaussian
from oracle import gp_oracle as orc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); walks = steps = 0; worst_w = worst_p = 0.0
while time.time() - t0 < budget:
    D = int(rng.randint(1, 4))
    ell = np.linspace(0.5, 1.0, D)
    gp = pygp_amd.ExactGP(Gaussian(0.15), pygp_amd.kernels.SE(1.0, ell), 0.0)
    spec0 = orc.se_spec(1.0, ell)
    B = int(rng.choice([2, 5, 30, 150]))
    base = gp.get_hyper()
    hypers = base + 0.15 * rng.randn(B, base.size)
    ens = HyperEnsemble(gp, hypers)
    total = int(rng.choice([rng.randint(3, 60), rng.randint(60, 400)]))
    X, y, Xs = recipes.synthetic(total, D, n_test=5, seed=int(rng.randint(10 ** 6)))
    lw = np.zeros(B) - np.log(B)
    before = np.zeros(B)
    at = 0
    while at < total:
        nxt = min(total, at + int(rng.choice([1, rng.randint(1, 8), rng.randint(8, 120)])))
        ens.add_data(X[at:nxt], y[at:nxt])
        at = nxt
        after = np.array([orc.exact_eval(spec0, th, X[:at], y[:at], grad=False) for th in hypers[:: max(1, B // 6)]])
        sel = np.arange(B)[:: max(1, B // 6)]
        full_after = np.asarray(ens._loglikes)                 # device values for all members
        assert np.max(np.abs(full_after[sel] - after) / np.maximum(np.abs(after), 1e-2)) <= 1e-8
        lw = lw + full_after - before
        lw -= logsumexp(lw)
        before = full_after
        worst_w = max(worst_w, float(np.max(np.abs(ens.logweights - lw))))
        assert np.allclose(ens.logweights, lw, rtol=0, atol=1e-9)
        if rng.randint(3) == 0:                                 # mixture posterior, weighted
            mu, s2 = ens.posterior(Xs)
            w = np.exp(lw)
            pm, ps = [], []
            for th in hypers:
                sb = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[1:-1])
                R, a = orc.exact_update(sb, th[0], th[-1], X[:at], y[:at])
                m_, s_ = orc.exact_posterior(sb, th[-1], X[:at], R, a, Xs)
                pm.append(m_); ps.append(s_)
            pm, ps = np.array(pm), np.array(ps)
            wm = w @ pm
            wv = w @ (ps + (pm - wm) ** 2)
            e = max(np.max(np.abs(mu - wm)), np.max(np.abs(s2 - wv)))
            worst_p = max(worst_p, float(e))
            assert e <= 1e-6, e
        if ens.ess() < B / 2 and rng.randint(2):
            idx = ens.resample(np.random.RandomState(int(rng.randint(10 ** 6))))
            hypers = hypers[idx]; lw = np.zeros(B) - np.log(B); before = before[idx]
            moved = hypers + 0.02 * rng.randn(*hypers.shape)      # a move step
            ens.set_hypers(moved); hypers = moved
            before = np.asarray(ens.loglikelihoods())
        steps += 1
    walks += 1
print('%d walks, %d steps; worst log-weight difference %.1e, mixture posterior %.1e' % (walks, steps, worst_w, worst_p))
print('smc walk ok')
