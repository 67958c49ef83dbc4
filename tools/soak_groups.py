#!/usr/bin/env python3
"""Randomised cross-check of the member-batched groups (pygp_amd/csrc/group.hip): random
sizes, dimensions, kernel families, batch lengths and value-only / with-gradients calls on
ONE handle; every batch against the same thetas evaluated one at a time (bit for bit), a
sample of members against the oracle, posteriors of a batch against the single model's, and
the device memory in use at the end of the run against its level after the first pass.
usage: soak_groups.py [seconds] [seed] [small]   (small: every kernel family of
tests/recipes.py and sizes from ONE point up -- the reference's own demo sizes)"""
import ctypes
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
from pygp_amd import _lib                          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
small = len(sys.argv) > 3 and sys.argv[3] == 'small'
hip = ctypes.CDLL('libamdhip64.so')


def used():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2 ** 20


FAMILIES = ['se_ard8', 'se_iso3', 'matern3_ard8', 'matern5_ard16', 'sum_se3', 'sum_se_per1',
            'rq_ard8', 'sum_prod3']
if small:
    FAMILIES = sorted(recipes.MID_CASES)
dev = _lib.Handle(0)
t0 = time.time()
n_batches = n_members = 0
worst = 0.0
while time.time() - t0 < budget:
    desc, D = recipes.MID_CASES[FAMILIES[rng.randint(len(FAMILIES))]]
    N = int(rng.choice([rng.randint(130, 700), rng.randint(700, 1400), rng.randint(1400, 2600),
                        rng.randint(2600, 4400)], p=[0.4, 0.3, 0.2, 0.1]))
    if small:
        N = int(rng.choice([rng.randint(1, 130), rng.randint(130, 700), rng.randint(700, 1400)],
                           p=[0.45, 0.4, 0.15]))
    B = int(rng.choice([2, 3, 5, 9, 17, 33, 70]))
    if N > 2600:
        B = min(B, 9)
    grad = bool(rng.randint(2))
    X, y, Xs = recipes.synthetic(N, D, n_test=int(rng.randint(1, 40)), seed=int(rng.randint(1000)))
    k = amd_kernel(desc)
    base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
    thetas = base + 0.05 * rng.randn(B, base.size)
    dev.set_data(X, y)
    out = dev.loglik_batch(k._kspec(), thetas, grad=grad)
    lZ = out[0] if grad else out
    for b in rng.permutation(B)[:4]:
        kb = k.copy(thetas[b][1:-1])
        one = dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], grad)
        if grad:
            assert one[0] == lZ[b] and np.array_equal(one[1], out[1][b]), (N, D, B, b, desc[0])
        else:
            assert one == lZ[b], (N, D, B, b, desc[0])
    b = int(rng.randint(B))
    spec = orc.spec_set_hyper(oracle_spec(desc), thetas[b][1:-1])
    R, a = orc.exact_update(spec, thetas[b][0], thetas[b][-1], X, y)
    want = orc.exact_loglik(spec, thetas[b][0], X, R, a, False)
    worst = max(worst, abs(lZ[b] - want) / max(abs(want), 1e-3))
    assert abs(lZ[b] - want) <= 1e-8 * abs(want) + (1e-11 if small else 0.0), (N, D, B, b, lZ[b], want)
    if rng.randint(3) == 0:
        mu, s2 = dev.posterior_batch(k._kspec(), thetas, Xs)
        kb = k.copy(thetas[b][1:-1])
        dev.exact_update(kb._kspec(), thetas[b][0], thetas[b][-1])
        m1, v1 = dev.exact_posterior(Xs)
        wm, ws = orc.exact_posterior(spec, thetas[b][-1], X, R, a, Xs)
        assert np.max(np.abs(mu[b] - wm)) <= 1e-6 and np.max(np.abs(s2[b] - ws)) <= 1e-6
        assert np.allclose(mu[b], m1, rtol=0, atol=1e-9) and np.allclose(s2[b], v1, rtol=0, atol=1e-9)
    n_batches += 1
    n_members += B
    if n_batches == 1:
        level = used()
print('%d batches, %d members in %.0f s; worst lZ error vs oracle %.1e; device memory in use '
      '%.0f MiB (after the first batch %.0f)' % (n_batches, n_members, time.time() - t0, worst,
                                                 used(), level))
print('soak ok')
