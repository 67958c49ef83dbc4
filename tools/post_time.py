#!/usr/bin/env python3
"""posterior latency vs number of test points, after a value-only update and
after a gradient evaluation (complete inverse). usage: post_time.py [N ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
D = 8
dev = _lib.Handle(0)
for N in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    X, y, _ = recipes.synthetic(N, D)
    th = recipes.theta_eval(D, 0)
    k = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
    dev.set_data(X, y)
    for mode in ('update', 'eval+grad'):
        if mode == 'update':
            dev.exact_update(k._kspec(), th[0], th[-1])
        else:
            dev.exact_eval(k._kspec(), th[0], th[-1], True)
        row = []
        for m in (1, 16, 128, 1024, 4096):
            Xs = np.random.RandomState(1).rand(m, D)
            dev.exact_posterior(Xs)
            t = []
            for _ in range(5):
                t0 = time.perf_counter(); dev.exact_posterior(Xs); t.append(time.perf_counter() - t0)
            row.append('m=%d: %.3f ms' % (m, np.median(t) * 1e3))
        print('N=%d after %-9s  ' % (N, mode) + '  '.join(row), flush=True)
        dev.enable_timing(True); dev.exact_posterior(Xs[:128]); tt = dev.timings(); dev.enable_timing(False)
        print('     m=128 stages: build %.3f solve %.3f ms' % (tt['posterior_build'], tt['posterior_solve']), flush=True)
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); dev.exact_posterior_grad(Xs[:128]); t.append(time.perf_counter() - t0)
        print('     posterior+input grads m=128: %.3f ms' % (np.median(t) * 1e3), flush=True)
