#!/usr/bin/env python3
"""Randomised walk over the model state machine of the reference (GP.add_data / set_hyper /
reset / copy / loglikelihood / posterior, /root/reference/pygp/inference/_base.py:59-186)
on the device models, every state checked against the oracle evaluated from scratch on the
data the model holds: first data vs incremental appends in chunks of random length (in
place, opening new 128-blocks, past the reserved capacity), hyperparameter changes after
appends, copies that keep appending, resets.
usage: soak_model.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes                                     # noqa: E402
from helpers import amd_kernel, oracle_spec        # noqa: E402
from oracle import gp_oracle as orc                # noqa: E402
import pygp_amd                                    # noqa: E402
from pygp_amd.likelihoods import Gaussian          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FAMILIES = sorted(recipes.MID_CASES)
t0 = time.time()
walks = checks = 0
worst = {'lZ': 0.0, 'dlZ': 0.0, 'mu': 0.0, 's2': 0.0, 'dmu': 0.0}


def check(gp, desc, X, y, Xs, tag):
    global checks
    h = gp.get_hyper()
    spec = orc.spec_set_hyper(oracle_spec(desc), h[1:-1])
    R, a = orc.exact_update(spec, h[0], h[-1], X, y)
    want_lZ, want_dlZ = orc.exact_loglik(spec, h[0], X, R, a, True)
    lZ0 = gp.loglikelihood()
    lZ, dlZ = gp.loglikelihood(True)
    e = max(abs(lZ0 - want_lZ), abs(lZ - want_lZ)) / max(abs(want_lZ), 1e-2)
    worst['lZ'] = max(worst['lZ'], e)
    assert e <= 1e-8, (tag, 'lZ', lZ0, lZ, want_lZ)
    e = np.max(np.abs(dlZ - want_dlZ)) / max(1.0, np.max(np.abs(want_dlZ)))
    worst['dlZ'] = max(worst['dlZ'], e)
    assert e <= 1e-6, (tag, 'dlZ', dlZ, want_dlZ)
    out = gp.posterior(Xs, grad=True)
    want = orc.exact_posterior_grad(spec, h[-1], X, R, a, Xs)
    for name, g, w in zip(('mu', 's2', 'dmu', 'dmu'), out, want):
        e = float(np.max(np.abs(g - w)))
        worst[name] = max(worst[name], e)
        assert e <= 1e-6, (tag, name, e)
    checks += 1


while time.time() - t0 < budget:
    name = FAMILIES[rng.randint(len(FAMILIES))]
    desc, D = recipes.MID_CASES[name]
    total = int(rng.choice([rng.randint(2, 140), rng.randint(140, 700), rng.randint(700, 1800)],
                           p=[0.4, 0.4, 0.2]))
    X, y, Xs = recipes.synthetic(total, D, n_test=int(rng.randint(1, 12)), seed=int(rng.randint(10 ** 6)))
    gp = pygp_amd.ExactGP(Gaussian(0.15), amd_kernel(desc), 0.05)
    at = 0
    step = 0
    while at < total and time.time() - t0 < budget:
        chunk = int(rng.choice([1, rng.randint(1, 6), rng.randint(1, 200), rng.randint(100, 900)]))
        nxt = min(total, at + chunk)
        gp.add_data(X[at:nxt], y[at:nxt])
        at = nxt
        step += 1
        op = rng.randint(8)
        if op == 0:                                   # hyperparameters move after appends
            gp.set_hyper(gp.get_hyper() + 0.03 * rng.randn(gp.nhyper))
        elif op == 1:                                 # a copy carries on, the original goes away
            gp = gp.copy()
        elif op == 2 and at > 3:                      # reset and re-add a prefix
            gp.reset()
            at = int(rng.randint(1, at))
            gp.add_data(X[:at], y[:at])
        if step <= 2 or rng.randint(3) == 0 or at == total:
            check(gp, desc, X[:at], y[:at], Xs, (name, total, at, step, op))
    assert gp.ndata == at
    walks += 1
print('%d walks, %d states checked in %.0f s; worst lZ %.1e (rel), dlZ %.1e, mu %.1e, s2 %.1e, '
      'dmu/ds2 %.1e' % (walks, checks, time.time() - t0, worst['lZ'], worst['dlZ'], worst['mu'],
                        worst['s2'], worst['dmu']))
print('soak ok')
