#!/usr/bin/env python3
"""Safe mode (gpx_set_safe_mode: diagonal blocks by recursion, no task-queue launches): single
evaluations, batches in groups and posteriors against the oracle at sizes that normally take
the leaf, one panel, a whole-matrix launch, the multi-block driver and the lock-step sweep; and
the automatic switch (`auto`, run with GPX_PANEL_TIMEOUT_US=10 in the environment): every
task-queue launch runs into its 10-us wait bound; a handle made with auto_safe_mode=True
warns -- the warning carries the original error --, switches and repeats the call, and says
so (safe_mode, safe_mode_switches); a default handle (`raise`) raises the error instead."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
from oracle import gp_oracle as orc
D = 3
ell = [0.6, 0.9, 1.2]
k = pygp_amd.kernels.SE(1.0, ell)
spec0 = orc.se_spec(1.0, ell)
auto = len(sys.argv) > 1 and sys.argv[1] == 'auto'
if len(sys.argv) > 1 and sys.argv[1] == 'raise':
    # default handle: the wait bound is an error, nothing switches silently
    dev = _lib.Handle(0)
    X, y, _ = recipes.synthetic(4000, D, seed=4000)
    dev.set_data(X, y)
    try:
        dev.exact_eval(k._kspec(), np.log(0.1), 0.05, False)
    except _lib.GpxError as e:
        assert 'timed out waiting' in str(e), e
        assert not dev.safe_mode and dev.safe_mode_switches == 0
        print('raised: %s' % e)
        print('raise ok')
        # (the aborted launch is over: the bound ended it)
        sys.exit(0)
    raise SystemExit('the launch did not run into the wait bound')
dev = _lib.Handle(0, auto_safe_mode=auto)
if not auto:
    dev.set_safe_mode(True)
    assert dev.safe_mode
worst = 0.0
warned = []
for N, B in [(100, 5), (700, 20), (1500, 3), (3000, 40), (4000, 2), (5000, 2)]:
    X, y, Xs = recipes.synthetic(N, D, n_test=11, seed=N)
    base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
    th = base + 0.05 * np.random.RandomState(N).randn(B, base.size)
    dev.set_data(X, y)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        lZ, dlZ = dev.loglik_batch(k._kspec(), th, grad=True)
        lZv = dev.loglik_batch(k._kspec(), th, grad=False)
        mu, s2 = dev.posterior_batch(k._kspec(), th, Xs)
        kb = k.copy(th[0][1:-1])
        one = dev.exact_eval(kb._kspec(), th[0][0], th[0][-1], True)
        onev = dev.exact_eval(kb._kspec(), th[0][0], th[0][-1], False)
    if wlist:
        print('  warning: %s' % wlist[0].message)
        warned.extend(str(w.message) for w in wlist)
    for b in (0, B - 1):
        sb = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[b][1:-1])
        R, a = orc.exact_update(sb, th[b][0], th[b][-1], X, y)
        want_lZ, want_dlZ = orc.exact_loglik(sb, th[b][0], X, R, a, True)
        wm, ws = orc.exact_posterior(sb, th[b][-1], X, R, a, Xs)
        e = max(abs(lZ[b] - want_lZ), abs(lZv[b] - want_lZ)) / abs(want_lZ)
        eg = np.max(np.abs(dlZ[b] - want_dlZ)) / max(1.0, np.max(np.abs(want_dlZ)))
        ep = max(np.max(np.abs(mu[b] - wm)), np.max(np.abs(s2[b] - ws)))
        worst = max(worst, e, eg, ep)
        assert e <= 1e-8 and eg <= 1e-7 and ep <= 1e-6, (N, B, b, e, eg, ep)
        if b == 0:
            assert abs(one[0] - want_lZ) <= 1e-8 * abs(want_lZ) and abs(onev - want_lZ) <= 1e-8 * abs(want_lZ)
    print('N=%d B=%d ok (safe mode %s)' % (N, B, dev.safe_mode), flush=True)
if auto:
    # exactly one switch, announced with the original error, visible on the handle
    assert dev.safe_mode and dev.safe_mode_switches == 1, (dev.safe_mode, dev.safe_mode_switches)
    assert len(warned) == 1 and 'timed out waiting' in warned[0] and 'safe mode' in warned[0], warned
    assert dev.batch_plan(4)['safe_mode'] is True
    print('worst error %.1e; auto ok' % worst)
else:
    assert dev.safe_mode_switches == 0 and dev.batch_plan(4)['safe_mode'] is True
    print('worst error %.1e; safe mode ok' % worst)
