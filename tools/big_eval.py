#!/usr/bin/env python3
"""One log-lik+grad evaluation at sizes past the metric config (a single GP on one
GPU: 3 x N^2 x 8 B of HBM), with the size-independent checks of
tests/test_gpu_gp.py::test_full_size_properties. usage: big_eval.py N [D]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
N = int(sys.argv[1]); D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, y, _ = recipes.synthetic(N, D)
gp = pygp_amd.BasicGP(0.1, 1.0, np.linspace(.5, 1.5, D))
t0 = time.perf_counter(); gp.add_data(X, y); t1 = time.perf_counter()
lZ, dlZ = gp.loglikelihood(True); t2 = time.perf_counter()
print('N=%d update %.1f ms, loglik+grad %.1f ms, lZ=%.10g' % (N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, lZ), flush=True)
th = gp.get_hyper()
t0 = time.perf_counter(); gp.set_hyper(th); lZ2, dlZ2 = gp.loglikelihood(True); t1 = time.perf_counter()
print('  second evaluation %.1f ms = %.1f TFLOP/s (N^3), deterministic: %s' %
      ((t1 - t0) * 1e3, float(N) ** 3 / (t1 - t0) * 1e-12, lZ2 == lZ and np.array_equal(dlZ, dlZ2)))
# interpolation: posterior at training inputs with the model's own noise
idx = np.random.RandomState(0).choice(N, 64, replace=False)
mu, s2 = gp.posterior(X[idx])
print('  posterior at 64 training inputs: max |mu - y| = %.3f, max s2 = %.3e, all finite: %s' %
      (np.max(np.abs(mu - y[idx])), s2.max(), bool(np.all(np.isfinite(dlZ)))))
# mean-gradient identity: dlZ/dmean = sum(alpha) = 1^T K^-1 (y - m)
eps = 1e-4
h = th.copy(); h[-1] += eps; gp.set_hyper(h); lp = gp.loglikelihood()
h[-1] -= 2 * eps; gp.set_hyper(h); lm = gp.loglikelihood()
print('  d lZ / d mean: analytic %.8g, central difference %.8g' % (dlZ[-1], (lp - lm) / (2 * eps)))
