#!/usr/bin/env python3
"""Four host threads, a handle each, random sizes (1 ... 2600 points) and batch lengths for
a while -- single evaluations, batches in groups, batch posteriors, all at once on one GPU:
every result against the same call made alone afterwards (bit for bit). Found at the end of
round 4: two panel launches from two streams can starve each other's spine workgroups
(panel.hip, "One panel launch at a time per device").
usage: soak_threads.py [seconds] [big]   (big: sizes up to 9 000 points -- the look-ahead
driver of single evaluations beside groups of other threads)"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
big = len(sys.argv) > 2 and sys.argv[2] == 'big'
D = 3
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
log = [[] for _ in range(4)]
err = []


def job(seed, j):
    rng = np.random.RandomState(seed * 1000 + j)
    N = int(rng.choice([rng.randint(1, 130), rng.randint(130, 1200), rng.randint(1200, 2600)]))
    B = int(rng.choice([1, 2, 5, 20, 70]))
    if big:
        N = int(rng.choice([rng.randint(1000, 4200), rng.randint(4200, 9000)]))
        B = int(rng.choice([1, 2, 3, 5, 9]))
    grad = bool(rng.randint(2))
    X, y, Xs = recipes.synthetic(N, D, n_test=5, seed=seed * 1000 + j)
    th = np.array([recipes.theta_sweep(D, b + j) for b in range(B)])
    return X, y, Xs, th, grad


def evaluate(dev, X, y, Xs, th, grad):
    dev.set_data(X, y)
    a = dev.loglik_batch(k._kspec(), th, grad=grad)
    kb = k.copy(th[0][1:-1])
    b = dev.exact_eval(kb._kspec(), th[0][0], th[0][-1], grad)
    c = dev.posterior_batch(k._kspec(), th, Xs)
    flat = lambda r: np.concatenate([np.ravel(np.asarray(v, float)) for v in (r if isinstance(r, tuple) else (r,))])
    return np.concatenate([flat(a), flat(b), flat(c)])


def worker(seed):
    try:
        dev = _lib.Handle(0)
        t0 = time.time()
        j = 0
        while time.time() - t0 < budget:
            log[seed].append(evaluate(dev, *job(seed, j)))
            j += 1
        dev.close()
    except Exception as e:                     # noqa: BLE001
        err.append((seed, repr(e)))


threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
for t in threads: t.start()
for t in threads: t.join()
assert not err, err
print('threads done: %d calls; replaying them alone' % sum(len(l) for l in log), flush=True)
dev = _lib.Handle(0)
n = nbad = 0
for seed in range(4):
    for j, got in enumerate(log[seed]):
        want = evaluate(dev, *job(seed, j))
        if not np.array_equal(got, want, equal_nan=True):
            X, y, Xs, th, grad = job(seed, j)
            bad = np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))
            rel = np.max(np.abs(got[bad] - want[bad]) / np.maximum(np.abs(want[bad]), 1e-300))
            print('MISMATCH thread %d call %d: N=%d B=%d grad=%s, %d of %d values differ, first at %d, '
                  'largest relative difference %.3e' % (seed, j, X.shape[0], th.shape[0], grad, bad.size,
                                                        got.size, bad[0], rel), flush=True)
            nbad += 1
        n += 1
        if n % 500 == 0:
            print('  %d replayed' % n, flush=True)
assert nbad == 0, '%d of %d calls differ' % (nbad, n)
print('%d calls from four threads, all bit-equal to the same calls alone' % n)
print('soak ok')
