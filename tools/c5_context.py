#!/usr/bin/env python3
"""What the config-5 build (fp32 SE+Periodic, N = 32768, D = 4) costs in which run
context (0.80 ms inside round 3's bench, 0.96 ms stand-alone and in
round 4's bench; the kernel's ISA is the same). Each line: context, ms per build for
reps = 10 (what tools/bench_configs.py times), sclk before / after from rocm-smi.
usage: python tools/c5_context.py [fresh|after_c4|all]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import recipes                                     # noqa: E402
from pygp_amd import _lib                          # noqa: E402


def sclk():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True,
                             text=True, timeout=20).stdout
        keep = [ln.split(':', 1)[1].strip() for ln in out.splitlines()
                if 'sclk' in ln or 'Power' in ln]
        return ' | '.join(keep)
    except Exception as e:                          # pragma: no cover
        return 'n/a (%s)' % e


def specs(D):
    hse = _lib.KSpecHolder(_lib.KIND_SE, False, D, np.r_[0.0, np.log(np.linspace(.5, 1.5, D))])
    hper = _lib.KSpecHolder(_lib.KIND_PERIODIC, False, D, np.r_[0.0, 0.0, np.log(0.7)])
    return hse, _lib.KSpecHolder(_lib.KIND_SUM, False, D, parts=[hse, hper])


def c5(dev, tag, reps=10, smi=True):
    N, D = 32768, 4
    X = np.random.RandomState(0).rand(N, D)
    dev.set_data(X, np.zeros(N))
    hse, hsum = specs(D)
    s0 = sclk() if smi else ''
    ms = dev.kernel_build_resident(hsum, np.float32, reps=reps)
    ms_se = dev.kernel_build_resident(hse, np.float32, reps=reps)
    s1 = sclk() if smi else ''
    print(json.dumps({'context': tag, 'reps': reps, 'se+periodic_fp32_ms': round(ms, 4),
                      'se_fp32_ms': round(ms_se, 4), 'smi_before': s0, 'smi_after': s1}),
          flush=True)
    return ms


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    dev = _lib.Handle(0)
    if what in ('fresh', 'all'):
        c5(dev, 'fresh handle, first call')
        c5(dev, 'fresh handle, second call')
        c5(dev, 'fresh handle, reps=300', reps=300)
        c5(dev, 'right after reps=300', smi=False)
        time.sleep(3.0)
        c5(dev, 'after 3 s idle', smi=False)
        # a busy GPU right in front: 100 ms of fp32 SE builds, then the timed call
        N, D = 32768, 4
        hse, hsum = specs(D)
        dev.kernel_build_resident(hse, np.float32, reps=120)
        c5(dev, 'right after 120 SE builds', smi=False)
        dev.kernel_build_resident(hsum, np.float32, reps=120)
        c5(dev, 'right after 120 SE+Periodic builds', smi=False)
    if what in ('after_c4', 'all'):
        import bench_configs
        r = bench_configs.run_c4(dev)
        print(json.dumps({'c4': [r['value_only_evals_per_s'], r['with_grad_evals_per_s']]}),
              flush=True)
        c5(dev, 'right after C4 on the same handle', smi=False)
        c5(dev, 'again', smi=True)
        dev2 = _lib.Handle(0)
        c5(dev2, 'second handle while the first holds the group workspaces')
        dev2.close()
        dev.close()
        dev = _lib.Handle(0)
        c5(dev, 'new handle after the first was closed')
    if what in ('after_mfma', 'all'):
        # fp64 products right in front (what C3 / C4 leave behind): clocks and power state
        N, D = 16384, 8
        X, y, _ = recipes.synthetic(N, D)
        dev.set_data(X, y)
        import pygp_amd
        k = pygp_amd.kernels.SE(1.0, np.ones(D))
        th = recipes.theta_eval(D, 0)
        for _ in range(4):
            dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
        c5(dev, 'right after four N=16384 evaluations', smi=False)
        c5(dev, 'again', smi=True)


if __name__ == '__main__':
    main()
