#!/usr/bin/env python3
"""The in-library multi-device entry with faked devices (GPX_MULTI_FAKE=1: ndev host threads
and handles, all on GPU 0) in one Python thread while another thread runs single evaluations
and batches on a handle of its own: every result against the same call alone."""
import os, sys, threading, time
import numpy as np
os.environ['GPX_MULTI_FAKE'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
D = 3
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
res = {'multi': [], 'lone': []}
err = []


def mjob(j):
    rng = np.random.RandomState(500 + j)
    N = int(rng.choice([rng.randint(50, 700), rng.randint(700, 2600), rng.randint(2600, 5000)]))
    B = int(rng.choice([1, 3, 8, 25]))
    ndev = int(rng.choice([2, 3, 8]))
    X, y, _ = recipes.synthetic(N, D, seed=j)
    th = np.array([recipes.theta_sweep(D, b + j) for b in range(B)])
    return X, y, th, ndev, bool(rng.randint(2))


def mcall(j):
    X, y, th, ndev, grad = mjob(j)
    out = _lib.loglik_batch_multi(k._kspec(), th, X, y, grad=grad, ndev=ndev)
    return np.concatenate([np.ravel(v) for v in (out if grad else (out,))])


def ljob(j):
    rng = np.random.RandomState(900 + j)
    N = int(rng.choice([rng.randint(50, 700), rng.randint(700, 2600), rng.randint(2600, 6000)]))
    X, y, _ = recipes.synthetic(N, D, seed=1000 + j)
    th = np.array([recipes.theta_sweep(D, b + j) for b in range(4)])
    return X, y, th, bool(rng.randint(2))


def lcall(dev, j):
    X, y, th, grad = ljob(j)
    dev.set_data(X, y)
    a = dev.exact_eval(k.copy(th[0][1:-1])._kspec(), th[0][0], th[0][-1], grad)
    b = dev.loglik_batch(k._kspec(), th, grad=grad)
    f = lambda r: np.concatenate([np.ravel(np.asarray(v, float)) for v in (r if isinstance(r, tuple) else (r,))])
    return np.concatenate([f(a), f(b)])


def worker(kind):
    try:
        t0 = time.time(); j = 0
        dev = _lib.Handle(0) if kind == 'lone' else None
        while time.time() - t0 < budget:
            res[kind].append(mcall(j) if kind == 'multi' else lcall(dev, j))
            j += 1
    except Exception as e:                 # noqa: BLE001
        err.append((kind, repr(e)))


ts = [threading.Thread(target=worker, args=(kd,)) for kd in ('multi', 'lone')]
for t in ts: t.start()
for t in ts: t.join()
assert not err, err
dev = _lib.Handle(0)
bad = 0
for j, got in enumerate(res['multi']):
    bad += not np.array_equal(got, mcall(j), equal_nan=True)
for j, got in enumerate(res['lone']):
    bad += not np.array_equal(got, lcall(dev, j), equal_nan=True)
print('%d multi-device calls, %d lone calls, %d differ from the same call alone' % (len(res['multi']), len(res['lone']), bad))
assert bad == 0
print('soak ok')
