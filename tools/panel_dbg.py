#!/usr/bin/env python3
"""Developer probe of the panel kernel: factor one n x n SPD matrix and compare
with LAPACK. usage: panel_dbg.py [n ...]"""
import os, sys
import numpy as np, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
for n in [int(a) for a in sys.argv[1:]] or [256]:
    rng = np.random.RandomState(n)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    A = (Q * np.logspace(0, 2, n)) @ Q.T
    print('n', n, 'start', flush=True)
    R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
    Rref = sla.cholesky(A)
    print('n', n, 'R err', np.abs(R - Rref).max(), 'W err', np.abs(Rinv @ Rref - np.eye(n)).max(),
          'Ainv err', np.abs(Ainv - np.linalg.inv(A)).max(), flush=True)
