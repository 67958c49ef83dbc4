#!/usr/bin/env python3
"""Throughput of the batched-theta path at the sizes its consumers use (the particle /
sample loops of /root/reference/pygp/meta/smc.py:86-126 and
/root/reference/pygp/learning/sampling.py:102-124 run hundreds of thetas on a few
hundred to a few thousand points): B thetas x N points through gpx_loglik_batch,
value-only and with gradients, each with its N^3/3 or N^3 roofline.
usage: python tools/batch_small.py [--b 256] [--sizes 512,1024,2048] [--d 8] [--reps 3]
       [--out FILE] [--check]   (--check: every member against the oracle, slow)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import recipes                                     # noqa: E402

PEAK_FP64_MFMA_TFLOPS = 78.6


def mfma_roof(flop, seconds):
    ach = flop / seconds * 1e-12
    return {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP64_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': ach / PEAK_FP64_MFMA_TFLOPS, 'flop': flop,
            'ms': seconds * 1e3}


def run_size(dev, N, D, B, reps=3):
    import pygp_amd
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev.set_data(X, y)
    spec = k._kspec()
    dev.loglik_batch(spec, thetas[:min(B, 8)], grad=True)          # warm-up: lists, contexts
    dev.loglik_batch(spec, thetas[:min(B, 8)], grad=False)
    out = {'config': 'batched sweep %d thetas x ExactGP SE-ARD N=%d D=%d, 1 GPU' % (B, N, D),
           'n': N, 'd': D, 'b': B}
    lz = {}
    for grad in (False, True):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = dev.loglik_batch(spec, thetas, grad=grad)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        key = 'with_grad' if grad else 'value_only'
        lz[key] = r[0] if grad else r
        out[key + '_s'] = t
        out[key + '_evals_per_s'] = B / t
        flop = B * float(N) ** 3 * (1.0 if grad else 1.0 / 3)
        out.setdefault('roofline', {})[
            '%d %s evals (B N^3%s flop)' % (B, 'loglik+grad' if grad else 'value-only',
                                           '' if grad else '/3')] = mfma_roof(flop, t)
    th = thetas[0]
    kk = k.copy(th[1:-1])
    for grad in (False, True):
        ts = []
        dev.exact_eval(kk._kspec(), th[0], th[-1], grad)
        for _ in range(5):
            t0 = time.perf_counter()
            r1 = dev.exact_eval(kk._kspec(), th[0], th[-1], grad)
            ts.append(time.perf_counter() - t0)
        out['one_eval_%s_ms' % ('with_grad' if grad else 'value_only')] = float(np.median(ts)) * 1e3
        one = r1[0] if grad else r1
        key = 'with_grad' if grad else 'value_only'
        out['member0_equals_single_' + key] = bool(one == lz[key][0])
    out['lZ0'] = float(lz['value_only'][0])
    out['value_only_equals_with_grad_lZ'] = bool(np.array_equal(lz['value_only'], lz['with_grad']))
    return out, thetas, lz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--b', type=int, default=256)
    ap.add_argument('--sizes', default='512,1024,2048')
    ap.add_argument('--d', type=int, default=8)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--out')
    ap.add_argument('--check', action='store_true')
    args = ap.parse_args()
    from pygp_amd import _lib
    dev = _lib.Handle(0)
    res = []
    for N in [int(s) for s in args.sizes.split(',')]:
        r, thetas, lz = run_size(dev, N, args.d, args.b, args.reps)
        if args.check:
            from oracle import gp_oracle as orc
            X, y, _ = recipes.synthetic(N, args.d)
            worst = 0.0
            for b in range(0, args.b, max(1, args.b // 8)):
                th = thetas[b]
                spec = orc.spec_set_hyper(orc.se_spec(1.0, np.ones(args.d)), th[1:-1])
                R, a = orc.exact_update(spec, th[0], th[-1], X, y)
                want = orc.exact_loglik(spec, th[0], X, R, a, False)
                worst = max(worst, abs(lz['value_only'][b] - want) / abs(want))
            r['max_rel_err_vs_oracle'] = worst
        print(json.dumps(r), flush=True)
        res.append(r)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
