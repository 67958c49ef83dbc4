#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by running the REFERENCE
(mwhoffman/pygp, mounted read-only at /root/reference) in the build container.

The reference is Python-2 era code with an un-vendored dependency (mwhutils),
so it is imported through a process-local, in-memory shim (nothing is written
into the reference tree, no bytecode is cached, nothing of the reference is
copied into this repository -- only inputs/outputs are stored).

Usage:  python tests/golden/make_golden.py [small] [mid] [c2] [c4] [metric] [c3]

  small   test-suite recipes of the reference (kernels 5x3 points, ExactGP 10
          points, demos/xy.npz flow)                       -> g_small.npz
  mid     N=200 kernels / ExactGP for every kernel family  -> g_mid.npz
  c2      BASELINE config 2  (N=4096  D=8  SE-ARD)         -> g_c2.npz
  c4      BASELINE config 4  (N=8192  D=8, first 2 thetas) -> g_c4.npz
  metric  metric config      (N=16384 D=8  SE-ARD)         -> g_metric.npz
  c3      BASELINE config 3  (N=16384 D=16 Matern-5/2)     -> g_c3.npz

Input recipes live in tests/recipes.py (shared with the tests, so that inputs
are regenerated from seeds instead of being stored).
"""

import os
import sys
import time
import types
import builtins
import itertools

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/ for recipes.py

import numpy as np
import scipy
import scipy.linalg as sla
import scipy.special

REF = '/root/reference'


def install_shim():
    """Make the py2-era reference importable on py3 / NumPy 2 (in memory)."""
    import abc
    builtins.xrange = range
    itertools.izip = zip
    import scipy.misc
    scipy.misc.logsumexp = scipy.special.logsumexp

    def rstate(rng=None):
        if rng is None:
            return np.random.mtrand._rand
        if isinstance(rng, np.random.RandomState):
            return rng
        return np.random.RandomState(rng)

    def chol_update(A, B, C, a, b):
        # block-append Cholesky (only reached through _updateinc; unused by
        # the goldens, present so that the import succeeds).
        n, m = A.shape[0], C.shape[0]
        B = sla.solve_triangular(A, B, trans=True)
        C = sla.cholesky(C - B.T @ B)
        R = np.zeros((n + m, n + m))
        R[:n, :n], R[:n, n:], R[n:, n:] = A, B, C
        c = sla.solve_triangular(C, b - B.T @ a, trans=True)
        return R, np.r_[a, c]

    m = types.ModuleType('mwhutils')
    m_abc = types.ModuleType('mwhutils.abc')
    m_abc.ABCMeta = abc.ABCMeta
    m_abc.abstractmethod = abc.abstractmethod
    m_abc.abstractclassmethod = lambda f: classmethod(abc.abstractmethod(f))
    m_rnd = types.ModuleType('mwhutils.random')
    m_rnd.rstate = rstate
    m_lin = types.ModuleType('mwhutils.linalg')
    m_lin.chol_update = chol_update
    m.abc, m.random, m.linalg = m_abc, m_rnd, m_lin
    for name, mod in [('mwhutils', m), ('mwhutils.abc', m_abc),
                      ('mwhutils.random', m_rnd), ('mwhutils.linalg', m_lin)]:
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    import pygp
    import pygp.kernels._combo as combo
    # NumPy 2: np.hstack no longer accepts a generator (_combo.py:91)
    combo.ComboKernel.get_hyper = \
        lambda self: np.hstack([p.get_hyper() for p in self._parts])
    return pygp


def meta():
    return dict(numpy=np.__version__, scipy=scipy.__version__,
                generated=time.strftime('%Y-%m-%d'),
                reference='mwhoffman/pygp @ /root/reference (py3 shim)')


def save(name, out):
    out['meta'] = np.array(repr(meta()))
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KiB' % (os.path.getsize(path) / 1024.))


def make_kernel(pk, desc):
    """Build a reference kernel from a recipe descriptor (tests/recipes.py)."""
    kind = desc[0]
    if kind == 'se':
        return pk.SE(*desc[1], **desc[2])
    if kind == 'matern':
        return pk.Matern(*desc[1], **desc[2])
    if kind == 'periodic':
        return pk.Periodic(*desc[1])
    if kind == 'rq':
        return pk.RQ(*desc[1], **desc[2])
    if kind == 'sum':
        parts = [make_kernel(pk, d) for d in desc[1]]
        k = parts[0]
        for p in parts[1:]:
            k = k + p
        return k
    if kind == 'product':
        parts = [make_kernel(pk, d) for d in desc[1]]
        k = parts[0]
        for p in parts[1:]:
            k = k * p
        return k
    raise ValueError(kind)


def gen_small(pygp):
    import recipes
    pk = pygp.kernels
    out = {}
    # G1: tests/test_kernels.py:90-95,161-225 recipes
    for name, desc in recipes.SMALL_KERNELS.items():
        k = make_kernel(pk, desc)
        x1, x2 = recipes.small_kernel_points(k.ndim)
        out['k.%s.hyper' % name] = k.get_hyper()
        out['k.%s.get12' % name] = k.get(x1, x2)
        out['k.%s.get11' % name] = k.get(x1)
        out['k.%s.grad12' % name] = np.array(list(k.grad(x1, x2)))
        out['k.%s.grad11' % name] = np.array(list(k.grad(x1)))
        out['k.%s.dget' % name] = k.dget(x1)
        out['k.%s.dgrad' % name] = np.array(list(k.dgrad(x1)))
        out['k.%s.gradx12' % name] = k.gradx(x1, x2)
        out['k.%s.grady12' % name] = k.grady(x1, x2)
    # G2: tests/test_inference.py:117-130,174-185 recipes
    for name, build in [
            ('exact', lambda: pygp.inference.ExactGP(
                pygp.likelihoods.Gaussian(1), pk.SE(1, 1, ndim=2), 0.0)),
            ('basic', lambda: pygp.inference.BasicGP(1, 1, 1, 0, ndim=2))]:
        gp = build()
        X, y, Xs, ys = recipes.inference_points(gp._kernel.ndim,
                                                gp._likelihood._logsigma)
        gp.add_data(X, y)
        lZ, dlZ = gp.loglikelihood(True)
        mu, s2 = gp.posterior(Xs)
        _, _, dmu, ds2 = gp.posterior(Xs, grad=True)
        out['gp.%s.dmu' % name], out['gp.%s.ds2' % name] = dmu, ds2
        out['gp.%s.X' % name], out['gp.%s.y' % name] = X, y
        out['gp.%s.Xs' % name], out['gp.%s.ys' % name] = Xs, ys
        out['gp.%s.hyper' % name] = gp.get_hyper()
        out['gp.%s.R' % name], out['gp.%s.a' % name] = gp._R, gp._a
        out['gp.%s.lZ' % name], out['gp.%s.dlZ' % name] = lZ, dlZ
        out['gp.%s.mu' % name], out['gp.%s.s2' % name] = mu, s2
        # full posterior and joint samples (exact.py:64-79, _base.py:143-178,
        # test_inference.py:81-83); the fake mwhutils.random.rstate maps an int
        # seed to RandomState(seed)
        fmu, fSigma = gp._full_posterior(gp._kernel.transform(Xs))
        out['gp.%s.full_mu' % name], out['gp.%s.full_Sigma' % name] = fmu, fSigma
        out['gp.%s.sample_latent' % name] = gp.sample(Xs, m=3, latent=True, rng=5)
        out['gp.%s.sample_noisy' % name] = gp.sample(Xs, m=2, latent=False, rng=6)
        out['gp.%s.sample_flat' % name] = gp.sample(Xs, rng=7)
        # hyper + 1 (test_inference.py:147-151)
        gp2 = gp.copy(gp.get_hyper() + 1)
        out['gp.%s.lZ_p1' % name] = gp2.loglikelihood()
        mu, s2 = gp2.posterior(Xs)
        out['gp.%s.mu_p1' % name], out['gp.%s.s2_p1' % name] = mu, s2
        # prior (no data) posterior (test_inference.py:37-41)
        gp3 = gp.copy()
        gp3.reset()
        mu, s2 = gp3.posterior(Xs)
        out['gp.%s.mu_prior' % name], out['gp.%s.s2_prior' % name] = mu, s2
        out['gp.%s.sample_prior' % name] = gp3.sample(Xs, m=2, rng=8)
    # G3: demos/basic.py:14-24 flow on demos/xy.npz (data file of the
    # reference's own test tests/test_learning.py:21-28)
    data = np.load(os.path.join(REF, 'pygp', 'demos', 'xy.npz'))
    X, y = data['X'], data['y']
    gp = pygp.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    lZ, dlZ = gp.loglikelihood(True)
    out['xy.X'], out['xy.y'] = X, y
    out['xy.hyper0'], out['xy.lZ0'], out['xy.dlZ0'] = gp.get_hyper(), lZ, dlZ
    xg = np.linspace(X.min(), X.max(), 50)[:, None]
    mu, s2 = gp.posterior(xg)
    out['xy.grid'], out['xy.mu0'], out['xy.s20'] = xg, mu, s2
    pygp.optimize(gp)
    out['xy.hyper_opt'], out['xy.lZ_opt'] = gp.get_hyper(), gp.loglikelihood()
    mu, s2 = gp.posterior(xg)
    out['xy.mu_opt'], out['xy.s2_opt'] = mu, s2
    gp = pygp.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    pygp.optimize(gp, {'sn': None})                # tests/test_learning.py:33
    out['xy.hyper_opt_fixsn'] = gp.get_hyper()
    # G4: learning.sample (sampling.py:80-146), the slice sampler over the hypers, with
    # the bounds of tests/recipes.py SAMPLE_BOUNDS as Uniform priors; raw chain, the
    # chain with 'mu' held fixed, and the mixture posterior of the raw=False models
    # (what meta/mcmc.py:75-93 computes from that list)
    from pygp.learning.sampling import sample
    from pygp.priors import Uniform
    gp = pygp.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    priors = dict((k_, Uniform(a, b)) for k_, (a, b) in recipes.SAMPLE_BOUNDS.items())
    hypers = sample(gp, priors, recipes.SAMPLE_N, raw=True, rng=recipes.SAMPLE_SEED)
    out['sample.hypers'] = hypers
    out['sample.final_hyper'] = gp.get_hyper()
    models = [gp.copy(h) for h in hypers]
    out['sample.loglikes'] = np.array([m.loglikelihood() for m in models])
    mu_, s2_ = [np.array(v) for v in zip(*[m.posterior(xg) for m in models])]
    mu = np.mean(mu_, axis=0)
    out['sample.mix_mu'] = mu
    out['sample.mix_s2'] = np.mean(s2_ + (mu_ - mu) ** 2, axis=0)
    gp = pygp.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    priors['mu'] = None
    out['sample.hypers_fixmu'] = sample(gp, priors, recipes.SAMPLE_N, raw=True,
                                        rng=recipes.SAMPLE_SEED + 1)
    save('g_small.npz', out)


def gen_mid(pygp):
    import recipes
    pk = pygp.kernels
    out = {}
    for name, (desc, D) in recipes.MID_CASES.items():
        k = make_kernel(pk, desc)
        X, y, Xs = recipes.synthetic(recipes.MID_N, D, n_test=32)
        gp = pygp.inference.ExactGP(pygp.likelihoods.Gaussian(0.1), k, 0.25)
        gp.add_data(X, y)
        lZ, dlZ = gp.loglikelihood(True)
        mu, s2 = gp.posterior(Xs)
        out['%s.hyper' % name] = gp.get_hyper()
        out['%s.K' % name] = k.get(X)
        out['%s.Ks' % name] = k.get(X, Xs)
        G = np.array(list(k.grad(X)))
        # one full gradient slice per hyper is large; keep a strided sample
        out['%s.grad_s' % name] = G[:, ::7, ::5]
        out['%s.grad_sum' % name] = G.sum(axis=(1, 2))
        out['%s.Rdiag' % name] = gp._R.diagonal().copy()
        out['%s.R_s' % name] = gp._R[::7, ::5]
        out['%s.a' % name] = gp._a
        out['%s.lZ' % name], out['%s.dlZ' % name] = lZ, dlZ
        out['%s.mu' % name], out['%s.s2' % name] = mu, s2
        _, _, dmu, ds2 = gp.posterior(Xs, grad=True)
        out['%s.dmu' % name], out['%s.ds2' % name] = dmu, ds2
    save('g_mid.npz', out)


def gen_big(pygp, tag):
    import recipes
    pk = pygp.kernels
    cfg = recipes.BIG_CASES[tag]
    out = {}
    X, y, Xs = recipes.synthetic(cfg['N'], cfg['D'], n_test=16)
    for i, theta in enumerate(cfg['thetas']()):
        k = make_kernel(pk, cfg['kernel'])
        gp = pygp.inference.ExactGP(pygp.likelihoods.Gaussian(1.0), k, 0.0)
        gp._X, gp._y = X, y            # attach without a throw-away _update
        t0 = time.time()
        gp.set_hyper(theta)            # -> _update (exact.py:50-55)
        t1 = time.time()
        lZ, dlZ = gp.loglikelihood(True)
        t2 = time.time()
        mu, s2 = gp.posterior(Xs)
        # input gradients of the posterior (exact.py:99-116) at full size
        _, _, dmu, ds2 = gp.posterior(Xs, grad=True)
        out['dmu%d' % i], out['ds2%d' % i] = dmu, ds2
        out['theta%d' % i] = theta
        out['lZ%d' % i], out['dlZ%d' % i] = lZ, dlZ
        out['mu%d' % i], out['s2%d' % i] = mu, s2
        out['a_head%d' % i] = gp._a[:64].copy()
        out['Rdiag_s%d' % i] = gp._R.diagonal()[::64].copy()
        out['t_update%d' % i], out['t_loglik%d' % i] = t1 - t0, t2 - t1
        print(tag, i, 'lZ=%.15g' % lZ, 'update %.1fs loglik %.1fs' %
              (t1 - t0, t2 - t1), flush=True)
        del gp
    save('g_%s.npz' % tag, out)


def gen_maunaloa(pygp):
    """The flow of pygp/demos/maunaloa.py:17-41 (SE + SE*Periodic + RQ + SE on the
    monthly CO2 series), evaluated by the reference; the demo's inputs are stored
    with the outputs because the data file does not travel."""
    import recipes
    pk = pygp.kernels
    data = np.loadtxt(os.path.join(REF, 'pygp', 'demos', 'maunaloa.txt')).flatten()
    data = np.array([(x, y) for x, y in enumerate(data) if y > -99])
    X = data[:, 0, None] / 12. + 1958
    y = data[:, 1]
    k = make_kernel(pk, recipes.MAUNALOA_KERNEL)
    gp = pygp.inference.ExactGP(pygp.likelihoods.Gaussian(0.2), k, y.mean())
    gp.add_data(X, y)
    lZ, dlZ = gp.loglikelihood(True)
    Xs = np.linspace(1958, 2020, 96)[:, None]
    mu, s2, dmu, ds2 = gp.posterior(Xs, grad=True)
    save('g_maunaloa.npz', dict(X=X, y=y, Xs=Xs, hyper=gp.get_hyper(), lZ=lZ, dlZ=dlZ,
                                mu=mu, s2=s2, dmu=dmu, ds2=ds2,
                                Rdiag=gp._R.diagonal().copy(), a=gp._a))
    print('maunaloa n=%d lZ=%.12g' % (len(y), lZ))


if __name__ == '__main__':
    what = sys.argv[1:] or ['small', 'mid']
    pygp = install_shim()
    for w in what:
        if w == 'small':
            gen_small(pygp)
        elif w == 'mid':
            gen_mid(pygp)
        elif w == 'maunaloa':
            gen_maunaloa(pygp)
        else:
            gen_big(pygp, w)
