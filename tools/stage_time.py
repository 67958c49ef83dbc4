import os, sys, numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import recipes, pygp_amd
from pygp_amd import _lib
for N in (2048, 4096):
    D=8; X,y,_=recipes.synthetic(N,D); dev=_lib.Handle(0); dev.set_data(X,y)
    k=pygp_amd.kernels.SE(1.0,np.ones(D)); dev.enable_timing(True)
    acc={}
    for i in range(8):
        th=recipes.theta_eval(D,i); dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
        if i>=2:
            for a,b in dev.timings().items(): acc[a]=acc.get(a,0)+b/6
    print(N, {a: round(b,3) for a,b in acc.items() if b>0}, 'sum', round(sum(acc.values()),3))
