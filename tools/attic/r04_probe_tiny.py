#!/usr/bin/env python3
"""Stage-by-stage probe of one small batch (N, D, B): prints a marker before every library
call; a call still running after 45 s dumps the Python stack and ends the process."""
import faulthandler, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
N, D, B = [int(a) for a in sys.argv[1:4]]
X, y, Xs = recipes.synthetic(N, D, n_test=6)
k = pygp_amd.kernels.SE(1.0, np.linspace(0.5, 1.5, D))
base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
thetas = base + 0.1 * np.random.RandomState(4000).randn(B, base.size)
dev = _lib.Handle(0)
dev.set_data(X, y)


def stage(name, f):
    print('stage', name, flush=True)
    faulthandler.dump_traceback_later(45, exit=True)
    t0 = time.perf_counter()
    r = f()
    faulthandler.cancel_dump_traceback_later()
    print('   done %.3f s' % (time.perf_counter() - t0), flush=True)
    return r


def single(b, grad):
    kb = k.copy(thetas[b][1:-1])
    return dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], grad)


stage('batch grad', lambda: dev.loglik_batch(k._kspec(), thetas, grad=True))
stage('batch value', lambda: dev.loglik_batch(k._kspec(), thetas, grad=False))
for b in range(B):
    stage('single grad %d' % b, lambda: single(b, True))
    stage('single value %d' % b, lambda: single(b, False))
stage('posterior batch', lambda: dev.posterior_batch(k._kspec(), thetas, Xs))
for b in (0, B - 1):
    kb = k.copy(thetas[b][1:-1])
    stage('update %d' % b, lambda: dev.exact_update(kb._kspec(), thetas[b][0], thetas[b][-1]))
    stage('posterior %d' % b, lambda: dev.exact_posterior(Xs))
print('all stages done', flush=True)
