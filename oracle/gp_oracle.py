"""
CPU oracle for the exact-GP hot path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy/SciPy restatement of the arithmetic of mwhoffman/pygp's
kernel-matrix + ExactGP path. It exists so that the HIP implementation in
``pygp_amd/csrc`` can be checked against something that follows the reference
call sequence step by step. It is NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it. ``pygp_amd`` never does, and raises if ``libgpx.so`` is missing.

Parity pinning: every function below is validated against the reference itself
(imported through the in-memory py3 shim of ``tests/golden/make_golden.py``)
by ``tests/test_oracle_golden.py`` on the committed ``tests/golden/*.npz``
vectors. The reference ships no literal golden outputs of its own
(SURVEY.md section 8c), so the goldens are outputs of the reference run in the
build container, with library versions recorded in each fixture.

A kernel is described by a plain dict ("spec"):

    {'kind': 'se',       'logsf': float, 'logell': float|array, 'iso': bool, 'ndim': int}
    {'kind': 'matern',   'logsf': ...,   'logell': ...,         'iso': bool, 'ndim': int, 'd': 1|3|5}
    {'kind': 'periodic', 'logsf': float, 'logell': float, 'logp': float}
    {'kind': 'sum',      'parts': [spec, ...]}
    {'kind': 'product',  'parts': [spec, ...]}

All citations are relative to /root/reference/.
"""

import numpy as np
import scipy.linalg as sla
import scipy.spatial.distance as ssd

__all__ = [
    'se_spec', 'matern_spec', 'periodic_spec', 'rq_spec', 'sum_spec', 'product_spec',
    'spec_nhyper', 'spec_get_hyper', 'spec_set_hyper',
    'kernel_get', 'kernel_grad', 'kernel_dget', 'kernel_dgrad',
    'kernel_gradx', 'kernel_grady',
    'exact_update', 'exact_loglik', 'exact_posterior', 'exact_posterior_grad', 'exact_full_posterior', 'gp_sample',
    'exact_eval',
]


# -- spec helpers ------------------------------------------------------------

def se_spec(sf, ell, ndim=None):
    """Mirror of SE.__init__ (pygp/kernels/se.py:25-38)."""
    logell = np.log(ell)
    iso = False
    nd = np.size(logell)
    if ndim is not None:
        if np.size(logell) != 1:
            raise ValueError('ndim only usable with scalar lengthscales')
        logell, iso, nd = float(logell), True, ndim
    return dict(kind='se', logsf=np.log(float(sf)), logell=logell, iso=iso,
                ndim=nd)


def matern_spec(sf, ell, d=3, ndim=None):
    """Mirror of Matern.__init__ (pygp/kernels/matern.py:25-42)."""
    spec = se_spec(sf, ell, ndim)
    if d not in (1, 3, 5):
        raise ValueError('d must be one of 1, 3, or 5')
    spec.update(kind='matern', d=d)
    return spec


def periodic_spec(sf, ell, p):
    """Mirror of Periodic.__init__ (pygp/kernels/periodic.py:31-36)."""
    return dict(kind='periodic', logsf=np.log(float(sf)),
                logell=np.log(float(ell)), logp=np.log(float(p)), ndim=1)


def rq_spec(sf, ell, alpha, ndim=None):
    """Mirror of RQ.__init__ (pygp/kernels/rq.py:22-37)."""
    spec = se_spec(sf, ell, ndim)
    spec.update(kind='rq', logalpha=np.log(float(alpha)))
    return spec


def sum_spec(*parts):
    """Mirror of the real SumKernel ctor (pygp/kernels/_real.py:86-94)."""
    flat = []
    for p in parts:
        flat += p['parts'] if p['kind'] == 'sum' else [p]
    if not all(p['ndim'] == flat[0]['ndim'] for p in flat):
        raise ValueError('cannot add mismatched kernels')
    return dict(kind='sum', parts=[_deepcopy_spec(p) for p in flat],
                ndim=flat[0]['ndim'])


def product_spec(*parts):
    """Mirror of the real ProductKernel ctor (pygp/kernels/_real.py:109-115) and
    of RealKernel.__mul__'s associativity (_real.py:35-36, _combo.py:151-161)."""
    flat = []
    for p in parts:
        flat += p['parts'] if p['kind'] == 'product' else [p]
    if not all(p['ndim'] == flat[0]['ndim'] for p in flat):
        raise ValueError('cannot multiply mismatched kernels')
    return dict(kind='product', parts=[_deepcopy_spec(p) for p in flat],
                ndim=flat[0]['ndim'])


def _product_but(values):
    """M[i] = product of every entry except the i-th (_combo.py:32-52)."""
    out = []
    for i in range(len(values)):
        acc = np.ones_like(values[i])
        for j, v in enumerate(values):
            if j != i:
                acc = acc * v
        out.append(acc)
    return out


_COMBO = ('sum', 'product')


def spec_nhyper(spec):
    if spec['kind'] in _COMBO:
        return sum(spec_nhyper(p) for p in spec['parts'])
    if spec['kind'] == 'periodic':
        return 3
    if spec['kind'] == 'rq':
        return 2 + np.size(spec['logell'])
    return 1 + np.size(spec['logell'])


def spec_get_hyper(spec):
    """se.py:46-47, matern.py:62-63, periodic.py:44-45, _combo.py:90-91."""
    if spec['kind'] in _COMBO:
        return np.hstack([spec_get_hyper(p) for p in spec['parts']])
    if spec['kind'] == 'periodic':
        return np.r_[spec['logsf'], spec['logell'], spec['logp']]
    if spec['kind'] == 'rq':                               # rq.py:47-48
        return np.r_[spec['logsf'], spec['logell'], spec['logalpha']]
    return np.r_[spec['logsf'], spec['logell']]


def spec_set_hyper(spec, hyper):
    """se.py:49-51, matern.py:65-67, periodic.py:47-50, _combo.py:93-98."""
    hyper = np.asarray(hyper, dtype=float)
    if spec['kind'] in _COMBO:
        a = 0
        for p in spec['parts']:
            b = a + spec_nhyper(p)
            spec_set_hyper(p, hyper[a:b])
            a = b
    elif spec['kind'] == 'periodic':
        spec['logsf'], spec['logell'], spec['logp'] = map(float, hyper[:3])
    elif spec['kind'] == 'rq':                             # rq.py:50-53
        spec['logsf'] = float(hyper[0])
        spec['logell'] = float(hyper[1]) if spec['iso'] else hyper[1:-1].copy()
        spec['logalpha'] = float(hyper[-1])
    else:
        spec['logsf'] = float(hyper[0])
        spec['logell'] = float(hyper[1]) if spec['iso'] else hyper[1:].copy()
    return spec


# -- distances (pygp/kernels/_distances.py) ----------------------------------

def _rescale(ell, X1, X2):
    # _distances.py:17-23
    return X1 / ell, (X2 / ell if X2 is not None else None)


def _sqdist(X1, X2=None):
    # _distances.py:35-41: direct-difference form, exact zeros on the diagonal
    return ssd.cdist(X1, X1 if X2 is None else X2, 'sqeuclidean')


def _sqdist_foreach(X1, X2=None):
    # _distances.py:44-52: one single-column cdist per input dimension
    X2 = X1 if X2 is None else X2
    for i in range(X1.shape[1]):
        yield ssd.cdist(X1[:, i, None], X2[:, i, None], 'sqeuclidean')


# -- kernels -----------------------------------------------------------------

def _matern_f(d, r):
    # matern.py:44-48
    return 1 if d == 1 else (1 + r if d == 3 else 1 + r * (1 + r / 3.))


def _matern_df(d, r):
    # matern.py:50-54
    return 1 if d == 1 else (r if d == 3 else r * (1 + r) / 3.)


def kernel_get(spec, X1, X2=None):
    """K(X1, X2). se.py:53-55, matern.py:69-74, periodic.py:53-59,
    _combo.py:103-108."""
    kind = spec['kind']
    if kind == 'sum':
        return sum(kernel_get(p, X1, X2) for p in spec['parts'])
    if kind == 'product':                                  # _combo.py:125-127
        out = 1
        for p in spec['parts']:
            out = out * kernel_get(p, X1, X2)
        return out
    if kind == 'se':
        A, B = _rescale(np.exp(spec['logell']), X1, X2)
        return np.exp(spec['logsf'] * 2 - _sqdist(A, B) / 2)
    if kind == 'matern':
        d = spec['d']
        A, B = _rescale(np.exp(spec['logell']) / np.sqrt(d), X1, X2)
        D = np.sqrt(_sqdist(A, B))
        return np.exp(spec['logsf'] * 2 - D) * _matern_f(d, D)
    if kind == 'periodic':
        sf2 = np.exp(spec['logsf'] * 2)
        ell = np.exp(spec['logell'])
        p = np.exp(spec['logp'])
        D = np.sqrt(_sqdist(X1, X2)) * np.pi / p
        return sf2 * np.exp(-2 * (np.sin(D) / ell) ** 2)
    if kind == 'rq':                                       # rq.py:54-61
        sf2 = np.exp(spec['logsf'] * 2)
        alpha = np.exp(spec['logalpha'])
        A, B = _rescale(np.exp(spec['logell']), X1, X2)
        return sf2 * (1 + 0.5 * _sqdist(A, B) / alpha) ** (-alpha)
    raise ValueError(kind)


def kernel_grad(spec, X1, X2=None):
    """Generator over dK/dtheta_i in the reference's hyper order.
    se.py:57-66, matern.py:76-90, periodic.py:61-74, _combo.py:114-116."""
    kind = spec['kind']
    if kind == 'sum':
        for p in spec['parts']:
            for g in kernel_grad(p, X1, X2):
                yield g
    elif kind == 'product':                                # _combo.py:133-138
        rest = _product_but([kernel_get(p, X1, X2) for p in spec['parts']])
        for Mi, p in zip(rest, spec['parts']):
            for g in kernel_grad(p, X1, X2):
                yield Mi * g
    elif kind == 'se':
        A, B = _rescale(np.exp(spec['logell']), X1, X2)
        D = _sqdist(A, B)
        K = np.exp(spec['logsf'] * 2 - D / 2)
        yield 2 * K
        if spec['iso']:
            yield K * D
        else:
            for Dd in _sqdist_foreach(A, B):
                yield K * Dd
    elif kind == 'matern':
        d = spec['d']
        A, B = _rescale(np.exp(spec['logell']) / np.sqrt(d), X1, X2)
        D = np.sqrt(_sqdist(A, B))
        S = np.exp(spec['logsf'] * 2 - D)
        K = S * _matern_f(d, D)
        M = S * _matern_df(d, D)
        yield 2 * K
        if spec['iso']:
            yield M * D
        else:
            for Dd in _sqdist_foreach(A, B):
                with np.errstate(invalid='ignore', divide='ignore'):
                    yield np.where(D < 1e-12, 0, M * Dd / D)
    elif kind == 'periodic':
        sf2 = np.exp(spec['logsf'] * 2)
        ell = np.exp(spec['logell'])
        p = np.exp(spec['logp'])
        D = np.sqrt(_sqdist(X1, X2)) * np.pi / p
        R = np.sin(D) / ell
        S = R ** 2
        E = 2 * sf2 * np.exp(-2 * S)
        yield E
        yield 2 * E * S
        yield 2 * E * R * D * np.cos(D) / ell
    elif kind == 'rq':                                     # rq.py:63-84
        sf2 = np.exp(spec['logsf'] * 2)
        alpha = np.exp(spec['logalpha'])
        A, B = _rescale(np.exp(spec['logell']), X1, X2)
        D = _sqdist(A, B)
        E = 1 + 0.5 * D / alpha
        K = sf2 * E ** (-alpha)
        M = K * D / E
        yield 2 * K
        if spec['iso']:
            yield M
        else:
            for Dd in _sqdist_foreach(A, B):
                yield K * Dd / E
        yield 0.5 * M - alpha * K * np.log(E)
    else:
        raise ValueError(kind)


def kernel_dget(spec, X):
    """k(x, x) per point. se.py:68-69, matern.py:92-93, periodic.py:76-77,
    _combo.py:110-112."""
    if spec['kind'] == 'sum':
        return sum(kernel_dget(p, X) for p in spec['parts'])
    if spec['kind'] == 'product':                          # _combo.py:129-131
        out = 1
        for p in spec['parts']:
            out = out * kernel_dget(p, X)
        return out
    return np.exp(spec['logsf'] * 2) * np.ones(len(X))


def kernel_dgrad(spec, X):
    """se.py:71-74, matern.py:95-98, periodic.py:79-82, _combo.py:118-120."""
    if spec['kind'] == 'sum':
        for p in spec['parts']:
            for g in kernel_dgrad(p, X):
                yield g
    elif spec['kind'] == 'product':                        # _combo.py:140-145
        rest = _product_but([kernel_dget(p, X) for p in spec['parts']])
        for Mi, p in zip(rest, spec['parts']):
            for g in kernel_dgrad(p, X):
                yield Mi * g
    else:
        yield 2 * kernel_dget(spec, X)
        for _ in range(spec_nhyper(spec) - 1):
            yield np.zeros(len(X))


def _diff(X1, X2=None):
    # _distances.py:26-32
    X2 = X1 if X2 is None else X2
    return X1[:, None, :] - X2[None, :, :]


def kernel_gradx(spec, X1, X2=None):
    """d k(x1, x2) / d x1, shape (n1, n2, d). se.py:76-83, matern.py:100-111,
    periodic.py:84-94, _real.py:96-97 (sum)."""
    kind = spec['kind']
    if kind == 'sum':
        return sum(kernel_gradx(p, X1, X2) for p in spec['parts'])
    if kind == 'product':                                  # _real.py:117-120
        rest = _product_but([kernel_get(p, X1, X2)[:, :, None] for p in spec['parts']])
        return sum(f * kernel_gradx(p, X1, X2) for f, p in zip(rest, spec['parts']))
    if kind == 'se':
        ell = np.exp(spec['logell'])
        A, B = _rescale(ell, X1, X2)
        D = _diff(A, B)
        K = np.exp(spec['logsf'] * 2 - np.sum(D ** 2, axis=-1) / 2)
        return -K[:, :, None] * D / ell
    if kind == 'matern':
        d = spec['d']
        ell = np.exp(spec['logell']) / np.sqrt(d)
        A, B = _rescale(ell, X1, X2)
        D1 = _diff(A, B)
        D = np.sqrt(np.sum(D1 ** 2, axis=-1))
        S = np.exp(spec['logsf'] * 2 - D)
        with np.errstate(invalid='ignore', divide='ignore'):
            M = np.where(D < 1e-12, 0, S * _matern_df(d, D) / D)
        return -M[:, :, None] * D1 / ell
    if kind == 'periodic':
        sf2 = np.exp(spec['logsf'] * 2)
        ell = np.exp(spec['logell'])
        p = np.exp(spec['logp'])
        D = _diff(X1, X2) * np.pi / p
        K = sf2 * np.exp(-2 * (np.sin(D) / ell) ** 2)
        return -2 * np.pi / ell ** 2 / p * K * np.sin(2 * D)
    if kind == 'rq':                                       # rq.py:93-107
        sf2 = np.exp(spec['logsf'] * 2)
        ell = np.exp(spec['logell'])
        alpha = np.exp(spec['logalpha'])
        A, B = _rescale(ell, X1, X2)
        D = _diff(A, B)
        E = 1 + np.sum(D ** 2, axis=-1) / 2 / alpha
        K = sf2 * E ** (-alpha)
        return -(K / E)[:, :, None] * D / ell
    raise ValueError(kind)


def kernel_grady(spec, X1, X2=None):
    """d k(x1, x2) / d x2 = -gradx (se.py:85-86, matern.py:113-114,
    periodic.py:96-97)."""
    return -kernel_gradx(spec, X1, X2)


# -- exact inference (pygp/inference/exact.py) -------------------------------

def exact_update(spec, log_sn, mean, X, y):
    """ExactGP._update, exact.py:50-55. Returns (R upper, a)."""
    sn2 = np.exp(log_sn * 2)                      # gaussian.py:36-39
    K = kernel_get(spec, X) + sn2 * np.eye(len(X))
    r = y - mean
    R = sla.cholesky(K)
    a = sla.solve_triangular(R, r, trans=True)
    return R, a


def exact_loglik(spec, log_sn, X, R, a, grad=False):
    """ExactGP.loglikelihood, exact.py:118-143."""
    n = len(a)
    lZ = -0.5 * np.inner(a, a)
    lZ -= 0.5 * np.log(2 * np.pi) * n
    lZ -= np.sum(np.log(R.diagonal()))
    if not grad:
        return lZ
    alpha = sla.solve_triangular(R, a, trans=False)
    Q = sla.cho_solve((R, False), np.eye(n))
    Q -= np.outer(alpha, alpha)
    sn2 = np.exp(log_sn * 2)
    dlZ = np.r_[
        -sn2 * np.trace(Q),
        [-0.5 * np.sum(Q * dK) for dK in kernel_grad(spec, X)],
        np.sum(alpha)]
    return lZ, dlZ


def exact_posterior(spec, mean, X, R, a, Xs):
    """ExactGP._marg_posterior(grad=False), exact.py:81-97."""
    mu = np.full(Xs.shape[0], float(mean))
    s2 = kernel_dget(spec, Xs)
    if X is not None:
        K = kernel_get(spec, X, Xs)
        RK = sla.solve_triangular(R, K, trans=True)
        mu += np.dot(RK.T, a)
        s2 -= np.sum(RK ** 2, axis=0)
    return mu, s2


def exact_full_posterior(spec, mean, X, R, a, Xs):
    """ExactGP._full_posterior, exact.py:64-79: mean vector and full covariance."""
    mu = np.full(Xs.shape[0], float(mean))
    Sigma = kernel_get(spec, Xs)
    if X is not None:
        K = kernel_get(spec, X, Xs)
        V = sla.solve_triangular(R, K, trans=True)
        mu = mu + np.dot(V.T, a)
        Sigma = Sigma - np.dot(V.T, V)
    return mu, Sigma


def gp_sample(mu, Sigma, log_sn, m=None, latent=True, rng=None):
    """GP.sample, _base.py:143-178 (rng: None -> NumPy's global state, int ->
    RandomState(int), RandomState -> itself, like mwhutils.random.rstate), with
    Gaussian.sample (gaussian.py:47-49) for latent=False."""
    if rng is None:
        rng = np.random.mtrand._rand
    elif not isinstance(rng, np.random.RandomState):
        rng = np.random.RandomState(rng)
    flatten = m is None
    m = 1 if flatten else m
    n = len(mu)
    Sigma = Sigma + 1e-10 * np.eye(n)
    f = mu[None] + np.dot(rng.normal(size=(m, n)), sla.cholesky(Sigma))
    if not latent:
        fr = f.ravel()
        f = (fr + rng.normal(size=len(fr), scale=np.exp(log_sn))).reshape(m, n)
    return f.ravel() if flatten else f


def exact_posterior_grad(spec, mean, X, R, a, Xs):
    """ExactGP._marg_posterior(grad=True), exact.py:81-116: also the
    derivatives of the predictive mean and variance w.r.t. the test inputs."""
    mu, s2 = exact_posterior(spec, mean, X, R, a, Xs)
    dmu = np.zeros_like(Xs)
    ds2 = np.zeros_like(Xs)
    if X is not None:
        n = X.shape[0]
        K = kernel_get(spec, X, Xs)
        RK = sla.solve_triangular(R, K, trans=True)
        dK = kernel_grady(spec, X, Xs)
        dK = dK.reshape(n, -1)
        RdK = sla.solve_triangular(R, dK, trans=True)
        dmu += np.dot(RdK.T, a).reshape(Xs.shape)
        RdK = np.rollaxis(np.reshape(RdK, (-1,) + Xs.shape), 2)
        ds2 -= 2 * np.sum(RdK * RK, axis=1).T
    return mu, s2, dmu, ds2


def exact_eval(spec, theta, X, y, grad=True):
    """One objective evaluation as optimize() does it
    (pygp/learning/optimization.py:54-59): set_hyper -> _update ->
    loglikelihood(grad). theta = [log sn | kernel hypers | mean]
    (pygp/inference/_base.py:91-108)."""
    theta = np.asarray(theta, dtype=float)
    spec = spec_set_hyper(_deepcopy_spec(spec), theta[1:-1])
    R, a = exact_update(spec, theta[0], theta[-1], X, y)
    return exact_loglik(spec, theta[0], X, R, a, grad)


def _deepcopy_spec(spec):
    out = dict(spec)
    if 'parts' in out:
        out['parts'] = [_deepcopy_spec(p) for p in out['parts']]
    if isinstance(out.get('logell'), np.ndarray):
        out['logell'] = out['logell'].copy()
    return out
