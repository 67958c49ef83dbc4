#!/usr/bin/env python3
"""Per-component relative error of dlZ against the reference goldens at the BASELINE
configs (the parity tests hold dlZ to 1e-7 of the LARGEST component; this prints what
every component actually does). usage: grad_err.py [metric c2 c3 c4]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes
from helpers import amd_kernel
import pygp_amd
from pygp_amd.likelihoods import Gaussian
for tag in sys.argv[1:] or ['metric', 'c3']:
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g_%s.npz' % tag))
    cfg = recipes.BIG_CASES[tag]
    X, y, _ = recipes.synthetic(cfg['N'], cfg['D'])
    i = 0
    while 'theta%d' % i in g.files:
        gp = pygp_amd.ExactGP(Gaussian(1.0), amd_kernel(cfg['kernel']), 0.0)
        gp._X, gp._y = X, y
        gp._data_changed()
        gp.set_hyper(g['theta%d' % i])
        lZ, dlZ = gp.loglikelihood(True)
        want = g['dlZ%d' % i]
        rel = np.abs(dlZ - want) / np.abs(want)
        print('%s theta%d N=%d: lZ rel err %.2e; max|dlZ| %.3g; per-component rel err of dlZ: max %.2e'
              % (tag, i, cfg['N'], abs(lZ - g['lZ%d' % i]) / abs(g['lZ%d' % i]), np.max(np.abs(want)),
                 rel.max()))
        print('   ', ' '.join('%.1e' % r for r in rel))
        i += 1
