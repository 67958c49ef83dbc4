#!/usr/bin/env python3
"""Two handles driven from two threads, compared with the single-threaded results
(the loop of tests/test_gpu_gp.py::test_two_handles_from_two_threads, repeated).
usage: thread_race.py [rounds]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
D = 3
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
jobs = [(500, 11), (777, 12)]
def run(slot, N, seed, out):
    X, y, _ = recipes.synthetic(N, D, seed=seed)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    thetas = np.array([recipes.theta_sweep(D, b + seed) for b in range(5)])
    res = []
    for _ in range(3):
        res.append(dev.loglik_batch(k._kspec(), thetas, grad=True))
    out[slot] = res
    dev.close()
want = {}
for i, (N, seed) in enumerate(jobs):
    run(i, N, seed, want)
bad = 0
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for r in range(rounds):
    got = {}
    ts = [threading.Thread(target=run, args=(i, N, seed, got)) for i, (N, seed) in enumerate(jobs)]
    for t in ts: t.start()
    for t in ts: t.join()
    for i in range(len(jobs)):
        for (l1, d1), (l2, d2) in zip(want[i], got[i]):
            if not (np.array_equal(l1, l2) and np.array_equal(d1, d2)):
                bad += 1
                print('round', r, 'job', i, 'lZ', l1, l2, flush=True)
print('mismatches', bad, 'of', rounds * 6)
