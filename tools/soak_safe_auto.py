#!/usr/bin/env python3
"""The automatic switch to safe mode: four threads with a handle each and the device-wide order
of panel launches switched OFF (GPX_PANEL_SERIAL=0) starve each other's launches, as two
processes on one GPU would. No call may fail: a handle made with auto_safe_mode=True whose
launch ran into the wait bound warns, switches to safe mode and repeats the call; results
within 1e-9 of the same calls alone. A developer soak that provokes device-side starvation on
purpose: NOT part of the GPU suite (since round 5 the suite covers the switch with one bounded
call, tools/check_safe_mode.py auto under GPX_PANEL_TIMEOUT_US=10)."""
import os, sys, threading, time, warnings
os.environ['GPX_PANEL_SERIAL'] = '0'
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd import _lib
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
D = 3
k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
log = [[] for _ in range(4)]
err, switched = [], []


def job(seed, j):
    rng = np.random.RandomState(seed * 1000 + j)
    N = int(rng.randint(1200, 4000))
    X, y, _ = recipes.synthetic(N, D, seed=seed * 1000 + j)
    return X, y, recipes.theta_sweep(D, j)


def evaluate(dev, X, y, th):
    dev.set_data(X, y)
    return dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], False)


def worker(seed):
    try:
        dev = _lib.Handle(0, auto_safe_mode=True)
        t0 = time.time(); j = 0
        while time.time() - t0 < budget:
            log[seed].append(evaluate(dev, *job(seed, j)))
            j += 1
        if dev.safe_mode_switches:
            switched.append(seed)
        dev.close()
    except Exception as e:                 # noqa: BLE001
        err.append((seed, repr(e)))


warnings.simplefilter('ignore')
ts = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
for t in ts: t.start()
for t in ts: t.join()
assert not err, err
dev = _lib.Handle(0)
worst = 0.0
for seed in range(4):
    for j, got in enumerate(log[seed]):
        want = evaluate(dev, *job(seed, j))
        worst = max(worst, abs(got - want) / abs(want))
print('%d calls, no failure; handles that switched to safe mode: %s; worst difference from the same call alone %.1e'
      % (sum(len(l) for l in log), switched, worst))
assert worst <= 1e-9
print('auto ok')
