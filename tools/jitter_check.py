#!/usr/bin/env python3
"""One line per operation with a digest of its results, over the sizes and entry points of
the path. Run it plain and under GPX_TEST_JITTER=<seed>[:<max_us>] (a pseudo-random delay
in front of every product, panel and build launch): the lines must not differ.
usage: jitter_check.py [quick]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes
import pygp_amd
from pygp_amd import _lib


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
dev = _lib.Handle(0)
sizes = (1024, 2048, 3001, 4096, 8192, 12288) + (() if quick else (16384,))
for N in sizes:
    for name, D, mk in (('se', 8, lambda D: pygp_amd.kernels.SE(1.0, np.ones(D))),
                        ('matern5', 16, lambda D: pygp_amd.kernels.Matern(1.0, np.ones(D), d=5))):
        if name == 'matern5' and N not in (3001, 12288):
            continue
        X, y, Xs = recipes.synthetic(N, D, n_test=64)
        dev.set_data(X, y)
        k = mk(D)
        th = recipes.theta_eval(D, 7)
        kk = k.copy(th[1:-1])
        lZ, dlZ = dev.exact_eval(kk._kspec(), th[0], th[-1], True)
        lZv = dev.exact_eval(kk._kspec(), th[0], th[-1], False)
        print('eval   %-8s N=%5d %s' % (name, N, dig(lZ, dlZ, lZv)), flush=True)
        thetas = np.array([recipes.theta_eval(D, 20 + b) for b in range(5 if N > 8192 else 9)])
        bl, bd = dev.loglik_batch(k._kspec(), thetas, grad=True)
        bv = dev.loglik_batch(k._kspec(), thetas, grad=False)
        print('batch  %-8s N=%5d %s' % (name, N, dig(bl, bd, bv)), flush=True)
        if N <= 8192:
            mu, s2 = dev.posterior_batch(k._kspec(), thetas[:4], Xs)
            dev.exact_update(kk._kspec(), th[0], th[-1])
            pg = dev.exact_posterior_grad(Xs[:7])
            print('post   %-8s N=%5d %s' % (name, N, dig(mu, s2, *pg)), flush=True)

# the in-library multi-device entry with faked devices (GPX_MULTI_FAKE=1 in the environment:
# three logical devices on this GPU, a host thread and a handle each, side by side) and an
# incremental update across a tile boundary
if os.environ.get('GPX_MULTI_FAKE') == '1':
    N, D = 4096, 8
    X, y, Xs = recipes.synthetic(N, D, n_test=16)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    thetas = np.array([recipes.theta_eval(D, 40 + b) for b in range(7)])
    ml, md = _lib.loglik_batch_multi(k._kspec(), thetas, X, y, grad=True, ndev=3)
    pm = _lib.posterior_batch_multi(k._kspec(), thetas[:5], Xs, X, y, grad=True, ndev=3)
    print('multi  se       N=%5d %s' % (N, dig(ml, md, *pm)), flush=True)
N, D = 2040, 8
X, y, Xs = recipes.synthetic(N + 40, D, n_test=16)
dev.set_data(X[:N], y[:N])
th = recipes.theta_eval(D, 9)
kk = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
dev.exact_update(kk._kspec(), th[0], th[-1])
outs = []
for i in range(0, 40, 8):
    dev.exact_append(X[N + i:N + i + 8], y[N + i:N + i + 8])
    outs += list(dev.exact_posterior(Xs))
print('append se       N=%5d %s' % (N, dig(*outs)), flush=True)
