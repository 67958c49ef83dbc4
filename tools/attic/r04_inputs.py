#!/usr/bin/env python3
"""Input forms the reference accepts through np.array(..., ndmin=2, dtype=float): lists,
float32, integers, Fortran order, strided views, 1-D inputs for 1-D kernels."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import recipes, pygp_amd
from pygp_amd.likelihoods import Gaussian
X, y, Xs = recipes.synthetic(300, 3, n_test=9)
def model(Xa, ya, Xsa):
    gp = pygp_amd.ExactGP(Gaussian(0.2), pygp_amd.kernels.SE(0.9, [0.5, 0.8, 1.1]), 0.1)
    gp.add_data(Xa, ya)
    lZ, dlZ = gp.loglikelihood(True)
    mu, s2, dmu, ds2 = gp.posterior(Xsa, grad=True)
    return np.r_[lZ, dlZ, mu, s2, dmu.ravel(), ds2.ravel()]
ref = model(X, y, Xs)
big = np.zeros((600, 6)); big[::2, ::2] = X
forms = {
    'lists': (X.tolist(), y.tolist(), Xs.tolist()),
    'fortran': (np.asfortranarray(X), y, np.asfortranarray(Xs)),
    'strided': (big[::2, ::2], np.repeat(y, 2)[::2], Xs[:, ::1]),
    'readonly': (X, y, Xs),
}
forms['readonly'][0].setflags(write=False)
for name, (a, b, c) in forms.items():
    got = model(a, b, c)
    assert np.array_equal(got, ref), name
    print(name, 'ok')
X32 = X.astype(np.float32)
got = model(X32, y.astype(np.float32), Xs.astype(np.float32))
want = model(X32.astype(float), y.astype(np.float32).astype(float), Xs.astype(np.float32).astype(float))
assert np.array_equal(got, want)
print('float32 ok')
Xi = (10 * X).astype(int); yi = (10 * y).astype(int)
assert np.array_equal(model(Xi, yi, Xs), model(Xi.astype(float), yi.astype(float), Xs))
print('integers ok')
# 1-D kernel, inputs as vectors (reference: ndmin=2 then transposed? _real.py:38-39)
gp = pygp_amd.ExactGP(Gaussian(0.2), pygp_amd.kernels.SE(0.9, 0.5, ndim=1), 0.0)
x1 = np.linspace(0, 1, 40)
try:
    gp.add_data(x1[:, None], np.sin(6 * x1))
    print('1-D column ok', gp.loglikelihood())
except Exception as e:
    print('1-D column failed', repr(e))
for bad in (np.array([[np.nan, 0.0, 1.0]]), np.array([[np.inf, 0.0, 1.0]])):
    gp2 = pygp_amd.ExactGP(Gaussian(0.2), pygp_amd.kernels.SE(0.9, [0.5, 0.8, 1.1]), 0.1)
    try:
        gp2.add_data(bad, np.array([1.0]))
        print('non-finite input accepted?!')
    except ValueError as e:
        print('non-finite input -> ValueError')
print('inputs ok')
