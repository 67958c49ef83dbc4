#!/usr/bin/env python3
"""CPU emulation (NumPy/SciPy) of a right-looking blocked Cholesky whose row panels come
from substitution, from the explicit inverse of the diagonal block, or from the explicit
inverse plus one refinement step: backward errors next to LAPACK (profiles/r03_cond_backward_err.txt)."""
import numpy as np, scipy.linalg as sla, time
def spd(n, seed, cond):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    ev = np.logspace(0, np.log10(cond), n)
    A=(Q * ev) @ Q.T
    return (A+A.T)/2
def blocked(A, nb, mode):
    n=A.shape[0]; A=A.copy(); R=np.zeros_like(A)
    for k in range(0,n,nb):
        e=min(n,k+nb)
        Rkk=sla.cholesky(A[k:e,k:e])
        R[k:e,k:e]=Rkk
        if e<n:
            if mode=='subst':
                R12=sla.solve_triangular(Rkk,A[k:e,e:],trans=True)
            else:
                W=sla.solve_triangular(Rkk,np.eye(e-k))   # explicit inverse
                R12=W.T@A[k:e,e:]
                if mode=='refine':
                    R12+=W.T@(A[k:e,e:]-Rkk.T@R12)
            R[k:e,e:]=R12
            A[e:,e:]-=R12.T@R12
    return R
n=2048
for cond in (1e6,1e10):
    A=spd(n,11,cond); nA=np.linalg.norm(A)
    Rl=sla.cholesky(A)
    print('cond %.0e lapack %.2e'%(cond,np.linalg.norm(Rl.T@Rl-A)/nA), end=' ')
    for mode in ('subst','expl','refine'):
        for nb in (128,512):
            R=blocked(A,nb,mode)
            print('%s/%d %.2e'%(mode,nb,np.linalg.norm(R.T@R-A)/nA), end=' ')
    print()
