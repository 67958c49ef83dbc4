#!/usr/bin/env python3
"""one-leaf check: where do R and R^-1 differ from LAPACK (by 16-blocks)?"""
import os, sys
import numpy as np, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
n = 128
rng = np.random.RandomState(1)
Q, _ = np.linalg.qr(rng.randn(n, n))
A = (Q * np.logspace(0, 2, n)) @ Q.T
try:
    R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
except Exception as e:
    print('failed:', e); sys.exit(0)
Rr = sla.cholesky(A)
E = np.abs(R - Rr).reshape(8, 16, 8, 16).max(axis=(1, 3))
print('R err by 16-blocks:\n', np.array2string(E, precision=1, max_line_width=200))
W = np.linalg.inv(Rr)
E = np.abs(Rinv - W).reshape(8, 16, 8, 16).max(axis=(1, 3))
print('W err by 16-blocks:\n', np.array2string(E, precision=1, max_line_width=200))
