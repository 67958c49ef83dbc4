#!/usr/bin/env python3
"""Timings of every BASELINE.json config on one MI355X (the headline metric
config is bench.py's job). Prints one JSON object per config.
usage: python tools/bench_configs.py [c2 c3 c4 c5]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes
import pygp_amd
from pygp_amd import _lib
from pygp_amd.likelihoods import Gaussian

which = sys.argv[1:] or ['c2', 'c3', 'c4', 'c5']
dev = _lib.Handle(0)


def timed(f, reps=3):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))


if 'c2' in which:        # ExactGP SE-ARD fp64, N=4096 D=8: build + Cholesky + posterior
    N, D, M = 4096, 8, 4096
    X, y, _ = recipes.synthetic(N, D)
    Xs = np.random.RandomState(1).rand(M, D)
    th = recipes.theta_eval(D, 0)
    k = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
    dev.set_data(X, y)
    dev.enable_timing(True)
    t_up = timed(lambda: dev.exact_update(k._kspec(), th[0], th[-1]))
    t_ev = timed(lambda: dev.exact_eval(k._kspec(), th[0], th[-1], True))
    dev.exact_update(k._kspec(), th[0], th[-1])
    t_po = timed(lambda: dev.exact_posterior(Xs))
    st = dev.timings()
    print(json.dumps({'config': 'C2 ExactGP SE-ARD fp64 N=4096 D=8', 'update_ms': t_up * 1e3,
                      'loglik_grad_eval_ms': t_ev * 1e3, 'posterior_4096pts_ms': t_po * 1e3,
                      'posterior_trsm_flop': float(N) * N * M,
                      'posterior_tflops': float(N) * N * M / t_po * 1e-12,
                      'posterior_stage_ms': {a: b for a, b in st.items() if 'posterior' in a}}))

if 'c3' in which:        # Matern-5/2 ARD fp64 N=16384 D=16: log-lik + 19-component gradient
    N, D = 16384, 16
    X, y, _ = recipes.synthetic(N, D)
    th = recipes.theta0(D, 2.0)
    k = pygp_amd.kernels.Matern(1.0, np.ones(D), d=5).copy(th[1:-1])
    dev.set_data(X, y)
    dev.enable_timing(True)
    out = {}
    def run():
        out['r'] = dev.exact_eval(k._kspec(), th[0], th[-1], True)
    t = timed(run)
    print(json.dumps({'config': 'C3 ExactGP Matern-5/2 ARD fp64 N=16384 D=16 loglik+grad',
                      'eval_ms': t * 1e3, 'evals_per_s': 1 / t, 'lZ': out['r'][0],
                      'tflops_N3': float(N) ** 3 / t * 1e-12,
                      'stage_ms': {a: b for a, b in dev.timings().items() if b > 0}}))

if 'c4' in which:        # 64 thetas x N=8192 D=8 on ONE GPU (the per-rank share is B/world)
    N, D, B = 8192, 8, 64
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev.set_data(X, y)
    dev.enable_timing(False)
    dev.loglik_batch(k._kspec(), thetas[:2], grad=False)
    t0 = time.perf_counter(); lZ = dev.loglik_batch(k._kspec(), thetas, grad=False)
    t_val = time.perf_counter() - t0
    t0 = time.perf_counter(); lZg, _ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    t_grad = time.perf_counter() - t0
    print(json.dumps({'config': 'C4 batched sweep 64 thetas x ExactGP SE-ARD N=8192 D=8, 1 GPU',
                      'value_only_s': t_val, 'value_only_evals_per_s': B / t_val,
                      'with_grad_s': t_grad, 'with_grad_evals_per_s': B / t_grad,
                      'lZ0': float(lZ[0]), 'lZ1': float(lZ[1])}))

if 'c5' in which:        # fp32 SE+Periodic build N=32768 D=4: GB/s against the HBM roof
    N, D = 32768, 4
    X = np.random.RandomState(0).rand(N, D)
    dev.set_data(X, np.zeros(N))
    hse = _lib.KSpecHolder(_lib.KIND_SE, False, D, np.r_[0.0, np.log(np.linspace(.5, 1.5, D))])
    hper = _lib.KSpecHolder(_lib.KIND_PERIODIC, False, D, np.r_[0.0, 0.0, np.log(0.7)])
    hsum = _lib.KSpecHolder(_lib.KIND_SUM, False, D, parts=[hse, hper])
    res = {}
    for name, spec in (('se+periodic', hsum), ('se', hse)):
        ms32 = dev.kernel_build_resident(spec, np.float32, reps=5)
        res[name] = {'fp32_ms': ms32, 'fp32_GBps': N * N * 4 / ms32 * 1e-6}
    ms64 = dev.kernel_build_resident(hse, np.float64, reps=3)
    res['se']['fp64_ms'] = ms64
    res['se']['fp64_GBps'] = N * N * 8 / ms64 * 1e-6
    print(json.dumps({'config': 'C5 kernel build N=32768 D=4 (full square, resident)',
                      'algorithmic_bytes_fp32': N * N * 4, 'hbm_peak_GBps': 8000, **res}))
