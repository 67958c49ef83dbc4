#!/usr/bin/env python3
"""Two PROCESSES on one GPU, each with value-only and with-gradient evaluations at N = 1200 ...
4000 for a while (whole-matrix panel launches of up to 250 workgroups from both): no call may
fail; a process whose launch is starved into the wait bound continues in safe mode."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import recipes, pygp_amd
from pygp_amd import _lib
from oracle import gp_oracle as orc
seed = int(sys.argv[1]); budget = float(sys.argv[2])
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, 3))
spec0 = orc.se_spec(1.0, np.linspace(.5, 1.5, 3))
dev = _lib.Handle(0)
t0 = time.time(); j = 0; nwarn = 0; worst = 0.0
while time.time() - t0 < budget:
    rng = np.random.RandomState(seed * 1000 + j)
    N = int(rng.randint(1200, 4000))
    X, y, _ = recipes.synthetic(N, 3, seed=seed * 1000 + j)
    th = recipes.theta_sweep(3, j)
    dev.set_data(X, y)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        lZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], bool(j & 1))
        nwarn += len(w)
    if j %% 40 == 0:
        sb = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[1:-1])
        R, a = orc.exact_update(sb, th[0], th[-1], X, y)
        want = orc.exact_loglik(sb, th[0], X, R, a, False)
        got = lZ[0] if isinstance(lZ, tuple) else lZ
        worst = max(worst, abs(got - want) / abs(want))
    j += 1
print('process %%d: %%d calls, %%d warnings, safe mode %%s, worst lZ error vs oracle %%.1e' %% (seed, j, nwarn, getattr(dev, '_safe_mode', False), worst), flush=True)
assert worst <= 1e-8
''' % (ROOT, os.path.join(ROOT, 'tests'))
budget = sys.argv[1] if len(sys.argv) > 1 else '20'
ps = [subprocess.Popen([sys.executable, '-c', code, str(i), budget]) for i in range(2)]
rcs = []
for p in ps:
    try:
        rcs.append(p.wait(timeout=float(budget) + 200))
    except subprocess.TimeoutExpired:
        p.kill(); rcs.append('timeout')
print('exit codes', rcs)
assert rcs == [0, 0]
print('two processes ok')
