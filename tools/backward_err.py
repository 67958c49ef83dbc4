#!/usr/bin/env python3
"""Backward error of the device factorisation vs LAPACK as cond(A) grows."""
import os, sys
import numpy as np, scipy.linalg as sla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.RandomState(0)
Q, _ = np.linalg.qr(rng.randn(n, n))
for cond in (1e2, 1e6, 1e8, 1e10, 1e12):
    A = (Q * np.logspace(0, np.log10(cond), n)) @ Q.T
    A = (A + A.T) / 2
    R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
    Rl = sla.cholesky(A)
    nA = np.linalg.norm(A)
    print('cond %.0e: ||R^T R - A||/||A||  device %.1e  lapack %.1e | ||R - R_lapack||/||R|| %.1e | '
          '||A Ainv - I|| device %.1e lapack %.1e'
          % (cond, np.linalg.norm(R.T @ R - A) / nA, np.linalg.norm(Rl.T @ Rl - A) / nA,
             np.linalg.norm(R - Rl) / np.linalg.norm(Rl),
             np.linalg.norm(A @ Ainv - np.eye(n)), np.linalg.norm(A @ sla.cho_solve((Rl, False), np.eye(n)) - np.eye(n))), flush=True)
