"""
Input recipes shared by the golden generator (tests/golden/make_golden.py),
the parity tests and bench.py. Inputs are always regenerated from seeds
(np.random.RandomState's legacy stream is frozen across NumPy versions), so the
fixtures only have to hold outputs.

Kernel descriptors are plain tuples so that the same table can build a
reference kernel (generator), an oracle spec (tests) or a pygp_amd kernel:
    ('se',       (sf, ell), {ndim})
    ('matern',   (sf, ell), {d, ndim})
    ('periodic', (sf, ell, p))
    ('rq',       (sf, ell, alpha), {ndim})
    ('sum',      [descriptor, ...])
    ('product',  [descriptor, ...])
"""

import numpy as np

# --- reference test-suite recipes -------------------------------------------

# /root/reference/tests/test_kernels.py:161-225 (the kernels on the hot path)
SMALL_KERNELS = {
    'se_ard': ('se', (0.8, [0.3, 0.4]), {}),
    'se_iso': ('se', (0.8, 0.3), {'ndim': 2}),
    'periodic': ('periodic', (0.5, 0.4, 0.3)),
    'matern_ard1': ('matern', (0.5, [0.4, 0.3]), {'d': 1}),
    'matern_ard3': ('matern', (0.5, [0.4, 0.3]), {'d': 3}),
    'matern_ard5': ('matern', (0.5, [0.4, 0.3]), {'d': 5}),
    'matern_iso1': ('matern', (0.5, 0.4), {'d': 1, 'ndim': 2}),
    'matern_iso3': ('matern', (0.5, 0.4), {'d': 3, 'ndim': 2}),
    'matern_iso5': ('matern', (0.5, 0.4), {'d': 5, 'ndim': 2}),
    'sum_se3': ('sum', [('se', (0.8, 0.3), {'ndim': 2}),
                        ('se', (0.1, 0.2), {'ndim': 2}),
                        ('se', (0.1, 0.2), {'ndim': 2})]),
    'sum_se_per': ('sum', [('se', (0.8, [0.3]), {}),
                           ('periodic', (0.5, 0.4, 0.3))]),
    'rq_ard': ('rq', (0.5, [0.4, 0.5], 0.3), {}),
    'rq_iso': ('rq', (0.5, 0.4, 0.3), {'ndim': 2}),
    # test_kernels.py:224-238
    'prod_se3': ('product', [('se', (0.8, 0.3), {'ndim': 2}),
                             ('se', (0.1, 0.2), {'ndim': 2}),
                             ('se', (0.1, 0.2), {'ndim': 2})]),
    'sum_prod_se': ('sum', [('product', [('se', (0.8, 0.3), {'ndim': 2}),
                                         ('se', (0.1, 0.2), {'ndim': 2})]),
                            ('product', [('se', (0.8, 0.3), {'ndim': 2}),
                                         ('se', (0.1, 0.2), {'ndim': 2})])]),
    'prod_mixed': ('product', [('matern', (0.7, [0.4, 0.6]), {'d': 3}),
                               ('rq', (0.9, 0.5, 0.8), {'ndim': 2})]),
    # a sum inside a product (_combo.py:123-146 accepts any parts): (A + B) C
    'prod_of_sum': ('product', [('sum', [('se', (0.8, 0.3), {'ndim': 2}),
                                         ('se', (0.4, [0.7, 0.5]), {})]),
                                ('matern', (0.6, [0.4, 0.5]), {'d': 3})]),
}


def small_kernel_points(ndim):
    """/root/reference/tests/test_kernels.py:93-95."""
    rng = np.random.RandomState(0)
    x1 = rng.rand(5, ndim)
    x2 = rng.rand(3, ndim)
    return x1, x2


def inference_points(ndim, logsigma):
    """/root/reference/tests/test_inference.py:118-130 (Gaussian.sample is
    f + rng.normal(size=len(f), scale=sigma), likelihoods/gaussian.py:47-49)."""
    rng = np.random.RandomState(1)
    sigma = np.exp(logsigma)
    X = rng.rand(10, ndim)
    y = rng.rand(10)
    y = y + rng.normal(size=10, scale=sigma)
    Xs = rng.rand(10, ndim)
    ys = rng.rand(10)
    ys = ys + rng.normal(size=10, scale=sigma)
    return X, y, Xs, ys


# --- synthetic recipe of SURVEY.md section 8(d) -----------------------------

def synthetic(N, D, n_test=0, seed=0):
    rng = np.random.RandomState(seed)
    X = rng.rand(N, D)
    y = np.sin(X.sum(1)) + 0.1 * rng.randn(N)
    Xs = np.random.RandomState(1).rand(n_test, D) if n_test else None
    return X, y, Xs


def theta0(D, ell_scale=1.0):
    """[log sn | log sf, log ell_1..D | mean] with sn=.1, sf=1,
    ell=linspace(.5,1.5,D)*ell_scale, mean=0."""
    return np.r_[np.log(0.1), np.log(1.0),
                 np.log(np.linspace(0.5, 1.5, D) * ell_scale), 0.0]


def theta_eval(D, i, ell_scale=1.0):
    """theta of bench evaluation i (perturbed so that nothing is cached)."""
    t = theta0(D, ell_scale)
    return t + 0.01 * np.random.RandomState(1000 + i).randn(t.size)


def theta_sweep(D, b):
    """theta of batched-sweep member b (BASELINE config 4)."""
    t = theta0(D)
    return t + 0.1 * np.random.RandomState(2000 + b).randn(t.size)


MID_N = 200
_L8 = list(np.linspace(0.5, 1.5, 8))
_L16 = list(np.linspace(0.5, 1.5, 16) * 2)
MID_CASES = {
    'se_ard8': (('se', (0.9, _L8), {}), 8),
    'se_iso3': (('se', (0.8, 0.7), {'ndim': 3}), 3),
    'matern1_ard8': (('matern', (0.9, _L8), {'d': 1}), 8),
    'matern3_ard8': (('matern', (0.9, _L8), {'d': 3}), 8),
    'matern5_ard16': (('matern', (1.1, _L16), {'d': 5}), 16),
    'matern3_iso2': (('matern', (0.7, 0.6), {'d': 3, 'ndim': 2}), 2),
    'periodic1': (('periodic', (0.5, 0.8, 0.7)), 1),
    'sum_se_per1': (('sum', [('se', (0.8, [0.3]), {}),
                             ('periodic', (0.5, 0.8, 0.7))]), 1),
    'sum_se3': (('sum', [('se', (0.8, 0.3), {'ndim': 2}),
                         ('se', (0.1, 0.2), {'ndim': 2}),
                         ('se', (0.1, 0.2), {'ndim': 2})]), 2),
    'rq_ard8': (('rq', (0.9, _L8, 1.7), {}), 8),
    'sum_rq_se2': (('sum', [('rq', (0.7, 0.6, 0.8), {'ndim': 2}),
                            ('se', (0.3, [0.4, 0.9]), {})]), 2),
    'prod_se_per1': (('product', [('se', (0.9, [1.5]), {}),
                                  ('periodic', (1.1, 0.8, 0.7))]), 1),
    'sum_prod3': (('sum', [('product', [('se', (0.8, [0.5, 0.9, 0.7]), {}),
                                        ('matern', (1.2, 0.8), {'d': 5, 'ndim': 3})]),
                           ('rq', (0.4, 0.6, 1.3), {'ndim': 3})]), 3),
    # (SE + RQ) Matern + SE: a sum inside a product inside a sum
    'prod_of_sum3': (('sum', [('product', [('sum', [('se', (0.8, [0.5, 0.9, 0.7]), {}),
                                                   ('rq', (0.4, 0.6, 1.3), {'ndim': 3})]),
                                          ('matern', (1.2, 0.8), {'d': 5, 'ndim': 3})]),
                              ('se', (0.2, 0.4), {'ndim': 3})]), 3),
}

# /root/reference/pygp/demos/maunaloa.py:27-31 (hypers near Rasmussen & Williams)
MAUNALOA_KERNEL = ('sum', [('se', (67, 66), {}),
                           ('product', [('se', (2.4, 90), {}),
                                        ('periodic', (1, 1, 1))]),
                           ('rq', (1.2, .66, 0.78), {}),
                           ('se', (0.15, 0.15), {})])

BIG_CASES = {
    # BASELINE.json configs[1]
    'c2': dict(N=4096, D=8, kernel=('se', (1.0, [1.0] * 8), {}),
               thetas=lambda: [theta_eval(8, 0)]),
    # BASELINE.json configs[3] (first two of the 64 thetas)
    'c4': dict(N=8192, D=8, kernel=('se', (1.0, [1.0] * 8), {}),
               thetas=lambda: [theta_sweep(8, 0), theta_sweep(8, 1)]),
    # BASELINE.json metric config
    'metric': dict(N=16384, D=8, kernel=('se', (1.0, [1.0] * 8), {}),
                   thetas=lambda: [theta0(8), theta_eval(8, 0)]),
    # BASELINE.json configs[2]
    'c3': dict(N=16384, D=16,
               kernel=('matern', (1.0, [1.0] * 16), {'d': 5}),
               thetas=lambda: [theta0(16, 2.0)]),
}


# learning.sample recipe (tests/golden/make_golden.py G4): Uniform priors on the
# BasicGP parameters of the xy.npz demo model, chain length and seed
SAMPLE_BOUNDS = {'sn': (0.01, 1.0), 'sf': (0.05, 5.0), 'ell': (0.01, 1.0), 'mu': (-2.0, 2.0)}
SAMPLE_N = 12
SAMPLE_SEED = 3
