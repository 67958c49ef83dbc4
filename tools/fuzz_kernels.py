#!/usr/bin/env python3
"""Wide-range differential check of the kernel builds: every family of tests/recipes.py with
hyperparameters drawn log-uniformly over six decades and inputs at three scales, K(X1, X2),
K(X), all hyperparameter slices of grad and the input gradients, device against the oracle
(/root/reference/pygp/kernels/*.py restated in oracle/gp_oracle.py). Error measure: absolute
difference over the largest magnitude of the array -- for gradient slices at least that of
K itself: a slice such as RQ's d/dlog(alpha) = K (D2 / 2E - alpha log E) cancels to 1e-16 of
K for tiny D2 / ell^2 in the reference's own formula, and what is left is rounding on both
sides.
usage: fuzz_kernels.py [seconds] [seed]"""
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

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
worst = {}
n = 0
while time.time() - t0 < budget:
    name = sorted(recipes.MID_CASES)[rng.randint(len(recipes.MID_CASES))]
    desc, D = recipes.MID_CASES[name]
    k = amd_kernel(desc)
    decades = rng.choice([1.0, 2.0, 3.0])
    h = k.get_hyper() + np.log(10.0) * decades * (2 * rng.rand(k.nhyper) - 1)
    k.set_hyper(h)
    spec = orc.spec_set_hyper(oracle_spec(desc), h)
    scale = 10.0 ** rng.randint(-2, 3)
    n1, n2 = int(rng.randint(1, 200)), int(rng.randint(1, 200))
    X1, X2 = scale * rng.rand(n1, D), scale * rng.rand(n2, D)
    if rng.randint(4) == 0:
        X2[: min(n1, n2)] = X1[: min(n1, n2)]      # coincident points (the r < 1e-12 guards)
    pairs = [('get', k.get(X1, X2), orc.kernel_get(spec, X1, X2)),
             ('get_self', k.get(X1), orc.kernel_get(spec, X1)),
             ('dget', k.dget(X1), orc.kernel_dget(spec, X1))]
    for i, (g, w) in enumerate(zip(k.grad(X1, X2), orc.kernel_grad(spec, X1, X2))):
        pairs.append(('grad%d' % i, g, w))
    for i, (g, w) in enumerate(zip(k.grad(X1), orc.kernel_grad(spec, X1))):
        pairs.append(('gradself%d' % i, g, w))
    try:
        pairs.append(('grady', k.grady(X1, X2), orc.kernel_grady(spec, X1, X2)))
    except (NotImplementedError, AttributeError):
        pass
    kmax = float(np.max(np.abs(pairs[0][2])))
    for tag, g, w in pairs:
        g, w = np.asarray(g, float), np.asarray(w, float)
        assert g.shape == w.shape, (name, tag, g.shape, w.shape)
        fin = np.isfinite(w)
        assert np.array_equal(np.isfinite(g), fin), (name, tag, 'finiteness', h)
        if not fin.any():
            continue
        ref = max(np.max(np.abs(w[fin])), kmax if tag.startswith('grad') else 0.0, 1e-300)
        e = np.max(np.abs(g[fin] - w[fin])) / ref
        if e > worst.get(name, (0,))[0]:
            worst[name] = (e, tag)
        assert e <= 1e-6, (name, tag, e, list(h), scale)
    n += 1
print('%d random kernels; worst relative-to-max error per family:' % n)
for name in sorted(worst):
    print('  %-16s %.1e (%s)' % (name, worst[name][0], worst[name][1]))
print('fuzz ok')
