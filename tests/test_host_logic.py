"""CPU-only tests of the host-side mirror of the reference interface: hyper
layouts, parameter names, constructor errors, copies. Recipes restated from
/root/reference/tests/test_kernels.py and tests/test_inference.py."""

import operator

import numpy as np
import numpy.testing as nt
import pytest

import recipes
from helpers import amd_kernel, oracle_spec
from oracle import gp_oracle as orc

import pygp_amd
import pygp_amd.kernels as pk
from pygp_amd.utils.models import get_params


@pytest.mark.parametrize('name', sorted(recipes.SMALL_KERNELS))
def test_kernel_hyper_layout(g_small, name):
    k = amd_kernel(recipes.SMALL_KERNELS[name])
    nt.assert_allclose(k.get_hyper(), g_small['k.%s.hyper' % name], rtol=0, atol=0)
    params = k._params()
    assert all(len(p) == 3 for p in params)
    assert sum(p[1] for p in params) == k.nhyper          # test_kernels.py:33-36
    h = k.get_hyper()
    k.set_hyper(h + 0.5)
    nt.assert_allclose(k.get_hyper(), h + 0.5)
    c = k.copy(h)
    nt.assert_allclose(c.get_hyper(), h)
    k.set_hyper(h)
    assert repr(k)
    x1, _ = recipes.small_kernel_points(k.ndim)
    nt.assert_allclose(k.dget(x1), g_small['k.%s.dget' % name])
    nt.assert_allclose(np.array(list(k.dgrad(x1))), g_small['k.%s.dgrad' % name])
    # the C description carries the same numbers
    spec = k._kspec()
    assert spec.c.nhyper == k.nhyper and spec.c.ndim == k.ndim
    assert orc.spec_nhyper(oracle_spec(recipes.SMALL_KERNELS[name])) == k.nhyper


def test_init_errors():
    # test_kernels.py:246-273
    with pytest.raises(ValueError):
        operator.add(pk.SE(1, 1, ndim=1), pk.SE(1, 1, ndim=2))
    with pytest.raises(ValueError):
        operator.mul(pk.SE(1, 1, ndim=1), pk.SE(1, 1, ndim=2))
    with pytest.raises(ValueError):
        pk.SE(1, [1, 1], ndim=1)
    with pytest.raises(ValueError):
        pk.Matern(1, [1, 1], ndim=1)
    with pytest.raises(ValueError):
        pk.Matern(1, 1, d=12)
    with pytest.raises(ValueError):
        pk.RQ(1, [1, 1], 1, ndim=1)
    # periodic is 1-d only (periodic.py:35) so SE(4-d) + Periodic is refused
    with pytest.raises(ValueError):
        operator.add(pk.SE(1, [1.0] * 4), pk.Periodic(1, 1, 1))
    # test_inference.py:215-230
    with pytest.raises(ValueError):
        pygp_amd.BasicGP(1, 1, 1, 0, 2, 'foo')
    with pytest.raises(ValueError):
        pygp_amd.ExactGP(object(), pk.SE(1, 1, ndim=2), 0)
    gp = pygp_amd.ExactGP(pygp_amd.likelihoods.Gaussian(1), pk.Periodic(1, 1, 1), 0)
    with pytest.raises(ValueError):
        pygp_amd.BasicGP.from_gp(gp)
    for name in ('se', 'matern1', 'matern3', 'matern5'):
        pygp_amd.BasicGP.from_gp(pygp_amd.BasicGP(1, 1, 1, 0, 2, name))


def test_gp_hyper_layout():
    gp = pygp_amd.BasicGP(sn=.1, sf=1, ell=.1, mu=0)       # demos/basic.py:22
    nt.assert_allclose(gp.get_hyper(), [np.log(.1), 0, np.log(.1), 0])
    assert [p[0] for p in gp._params()] == ['sn', 'sf', 'ell', 'mu']
    assert gp.nhyper == 4 and gp.ndata == 0 and gp.data == (None, None)
    names = [n for n, _, _ in get_params(gp)]
    assert names == ['sn', 'sf', 'ell', 'mu']
    gp2 = pygp_amd.ExactGP(pygp_amd.likelihoods.Gaussian(1), pk.SE(1, 1, ndim=2), 0.0)
    assert [p[0] for p in gp2._params()] == ['like.sigma', 'kern.sf', 'kern.ell',
                                             'mean']
    # no data: set_hyper only stores, copy is independent, prior posterior works
    gp2.set_hyper(gp2.get_hyper() + 1)
    c = gp2.copy()
    c.set_hyper(c.get_hyper() - 1)
    nt.assert_allclose(gp2.get_hyper() - 1, c.get_hyper())
    mu, s2 = c.posterior(np.random.RandomState(0).rand(4, 2))
    nt.assert_allclose(mu, 0.0)
    nt.assert_allclose(s2, 1.0)
    assert repr(gp2) and repr(gp)
    with pytest.raises(ValueError):
        gp2.loglikelihood()


def test_sum_kernel_flattening():
    k = pk.SE(0.8, 0.3, ndim=2) + pk.SE(0.1, 0.2, ndim=2) + pk.SE(0.1, 0.2, ndim=2)
    assert len(k._parts) == 3 and k.nhyper == 6
    assert [p[0] for p in k._params()] == ['part0.sf', 'part0.ell', 'part1.sf',
                                           'part1.ell', 'part2.sf', 'part2.ell']
    spec = k._kspec()
    assert spec.c.nparts == 3 and spec.c.nhyper == 6
    # parts are copies (ComboKernel.__init__, _combo.py:61-63)
    a = pk.SE(1, 1, ndim=2)
    s = a + a
    s.set_hyper(np.arange(4.0))
    nt.assert_allclose(a.get_hyper(), [0, 0])


def test_product_kernel_flattening():
    a, b = pk.SE(0.8, 0.3, ndim=2), pk.SE(0.1, 0.2, ndim=2)
    k = a * b * b
    assert isinstance(k, pk.ProductKernel) and len(k._parts) == 3
    k = a * b + a * b                                      # test_kernels.py:232-238
    assert isinstance(k, pk.SumKernel) and len(k._parts) == 2 and k.nhyper == 8
    # flat numbering of the leaves through the nesting (_combo.py:73-88)
    assert [p[0] for p in k._params()] == [
        'part%d.%s' % (i, n) for i in range(4) for n in ('sf', 'ell')]
    spec = k._kspec()
    assert spec.c.nparts == 2 and spec.c.nhyper == 8
    assert spec.parts[0].c.kind == pygp_amd._lib.KIND_PRODUCT
    h = np.arange(8.0) / 10
    k.set_hyper(h)
    nt.assert_allclose(k.get_hyper(), h)
    nt.assert_allclose(k._parts[1].get_hyper(), h[4:])
    # a sum inside a product (_combo.py:123-146 accepts any parts): one hyper vector,
    # parts in tree order; dget / dgrad follow the product rule on the host
    k = operator.mul(a + b, a)
    assert isinstance(k, pk.ProductKernel) and k.nhyper == 6
    assert isinstance(k._parts[0], pk.SumKernel)
    X = np.random.RandomState(0).rand(4, 2)
    nt.assert_allclose(k.dget(X), (0.8 ** 2 + 0.1 ** 2) * 0.8 ** 2)
    dg = list(k.dgrad(X))
    assert len(dg) == 6
    nt.assert_allclose(dg[0], 2 * 0.8 ** 2 * 0.8 ** 2)        # d/dlog sf_a of (a + b) a
    nt.assert_allclose(dg[4], 2 * (0.8 ** 2 + 0.1 ** 2) * 0.8 ** 2)
    spec = k._kspec()
    assert spec.c.kind == pygp_amd._lib.KIND_PRODUCT
    assert spec.parts[0].c.kind == pygp_amd._lib.KIND_SUM and spec.c.nhyper == 6


def test_non_finite_inputs_raise_before_the_device():
    """sla.cholesky(check_finite=True) at exact.py:54 raises ValueError for NaN/inf;
    the mirror refuses them on the host (no GPU needed to get there)."""
    gp = pygp_amd.BasicGP(0.1, 1.0, 0.5, ndim=2)
    X = np.random.RandomState(0).rand(6, 2)
    y = np.arange(6.0)
    Xbad = X.copy()
    Xbad[3, 1] = np.nan
    with pytest.raises(ValueError):
        gp.add_data(Xbad, y)
    gp = pygp_amd.BasicGP(0.1, 1.0, 0.5, ndim=2)
    ybad = y.copy()
    ybad[0] = np.inf
    with pytest.raises(ValueError):
        gp.add_data(X, ybad)
    gp = pygp_amd.BasicGP(0.1, 1.0, 0.5, ndim=2)
    gp._X, gp._y = X, y                      # data attached, nothing uploaded yet
    h = gp.get_hyper()
    h[1] = np.nan
    with pytest.raises(ValueError):
        gp.set_hyper(h)


class _Uniform(object):
    """Box prior with the logprior() of the reference's priors.Uniform
    (pygp/priors/priors.py:38-44); the prior classes themselves are out of scope."""
    def __init__(self, a, b):
        self.a, self.b = np.atleast_1d(a).astype(float), np.atleast_1d(b).astype(float)

    def logprior(self, theta):
        theta = np.atleast_1d(theta)
        return 0.0 if np.all((theta >= self.a) & (theta <= self.b)) else -np.inf


class _OracleGP(object):
    """Host-only stand-in with the GP interface sample() touches (nhyper, _params,
    get_hyper, set_hyper, loglikelihood), evaluated by the oracle."""
    def __init__(self, X, y, hyper):
        self.X, self.y, self.h = X, y, np.array(hyper, dtype=float)
        self.nhyper = len(self.h)
        self.spec = orc.se_spec(1.0, np.ones(X.shape[1]))

    def _params(self):
        return [('sn', 1, True), ('sf', 1, True), ('ell', self.X.shape[1], True),
                ('mu', 1, False)]

    def get_hyper(self):
        return self.h.copy()

    def set_hyper(self, h):
        self.h = np.array(h, dtype=float)

    def loglikelihood(self):
        return orc.exact_eval(self.spec, self.h, self.X, self.y, grad=False)


def test_slice_sampler_reproduces_the_reference_chain(g_small):
    """learning.sample on a host-only model: the chain of the reference's sampler
    (sampling.py:24-146) for the same seed, priors and data, also with a frozen
    block (priors[name] = None)."""
    import recipes
    from pygp_amd.learning import sample
    X, y = g_small['xy.X'], g_small['xy.y']
    h0 = np.r_[np.log(.1), np.log(1.), np.log(.1), 0.0]
    priors = dict((k, _Uniform(*v)) for k, v in recipes.SAMPLE_BOUNDS.items())
    gp = _OracleGP(X, y, h0)
    hypers = sample(gp, priors, recipes.SAMPLE_N, rng=recipes.SAMPLE_SEED)
    nt.assert_allclose(hypers, g_small['sample.hypers'], rtol=1e-9, atol=1e-9)
    nt.assert_allclose(gp.get_hyper(), g_small['sample.final_hyper'], rtol=1e-9, atol=1e-9)
    gp = _OracleGP(X, y, h0)
    priors['mu'] = None
    hypers = sample(gp, priors, recipes.SAMPLE_N, rng=recipes.SAMPLE_SEED + 1)
    nt.assert_allclose(hypers, g_small['sample.hypers_fixmu'], rtol=1e-9, atol=1e-9)
    assert np.all(hypers[:, -1] == 0.0)


def test_multi_device_data_goes_down_once_per_data_set(monkeypatch):
    """HyperEnsemble(ndev=N) hands X, y to the in-library multi-device entries only when
    the devices do not hold its current data set already: once per data set, again
    after add_data, again after another ensemble used the (process-wide) handles, and
    again when more devices are asked for than hold the data."""
    from pygp_amd import _lib
    from pygp_amd.meta import HyperEnsemble
    calls = []

    def fake_loglik(spec, thetas, X, y, grad, ndev):
        calls.append(('L', X is not None, ndev))
        return np.zeros(len(thetas))

    def fake_post(spec, thetas, Xs, X, y, grad, ndev):
        calls.append(('P', X is not None, ndev))
        m = len(Xs)
        return np.zeros((len(thetas), m)), np.ones((len(thetas), m))

    monkeypatch.setattr(_lib, '_loglik_batch_multi', fake_loglik)
    monkeypatch.setattr(_lib, '_posterior_batch_multi', fake_post)
    monkeypatch.setattr(_lib, '_multi_resident', (None, 0))
    rng = np.random.RandomState(0)
    gp = pygp_amd.BasicGP(.1, 1., [1., 1.])
    gp._X, gp._y = rng.rand(6, 2), rng.rand(6)          # attach without a device update
    H = np.tile(gp.get_hyper(), (3, 1))
    a = HyperEnsemble(gp, H, ndev=2)
    a.loglikelihoods()
    a.loglikelihoods()
    a.posterior(rng.rand(4, 2))
    assert calls == [('L', True, 2), ('L', False, 2), ('P', False, 2)]
    b = HyperEnsemble(gp, H, ndev=2)                     # another owner of the handles
    b.loglikelihoods()
    a.loglikelihoods()
    assert calls[3:] == [('L', True, 2), ('L', True, 2)]
    a._ndev = 4                                          # more devices than hold the data
    a.loglikelihoods()
    a.loglikelihoods()
    assert calls[5:] == [('L', True, 4), ('L', False, 4)]
    del calls[:]
    a._model._data_changed = lambda: None
    a.add_data(rng.rand(2, 2), rng.rand(2))              # new data set: one upload
    a.loglikelihoods()
    assert [c[1] for c in calls] == [True, False]
    # callers without a token (bench.py, tests) always upload what they pass
    _lib.loglik_batch_multi(None, H, gp._X, gp._y, ndev=4)
    a.loglikelihoods()
    assert [c[1] for c in calls[2:]] == [True, True]


def test_residency_record_after_a_failed_call_and_for_copies(monkeypatch):
    """ A multi-device call that raises leaves the residency record empty -- the C
    side drops its resident data on an error path, and a stale record would make every later
    call pass X = NULL and fail with "no data is resident" -- and a deep copy of an ensemble
    is another owner: its own token, so that two ensembles whose data diverge never evaluate
    on each other's resident copy."""
    import copy
    from pygp_amd import _lib
    from pygp_amd.meta import HyperEnsemble
    calls = []
    state = {'fail': False}

    def fake_loglik(spec, thetas, X, y, grad, ndev):
        calls.append(X is not None)
        if state['fail']:
            raise _lib.GpxError('device lost')
        return np.zeros(len(thetas))

    monkeypatch.setattr(_lib, '_loglik_batch_multi', fake_loglik)
    monkeypatch.setattr(_lib, '_multi_resident', (None, 0))
    rng = np.random.RandomState(0)
    gp = pygp_amd.BasicGP(.1, 1., [1., 1.])
    gp._X, gp._y = rng.rand(6, 2), rng.rand(6)
    H = np.tile(gp.get_hyper(), (3, 1))
    a = HyperEnsemble(gp, H, ndev=2)
    a.loglikelihoods()
    a.loglikelihoods()
    assert calls == [True, False]
    state['fail'] = True
    with pytest.raises(_lib.GpxError):
        a.loglikelihoods()
    state['fail'] = False
    a.loglikelihoods()                                  # uploads again: the record was dropped
    assert calls[2:] == [False, True]
    b = copy.deepcopy(a)
    assert b._multi_token() != a._multi_token()
    b.loglikelihoods()                                  # another owner: its data go down
    a.loglikelihoods()
    assert calls[4:] == [True, True]


def test_the_switch_to_safe_mode_is_opt_in_and_visible():
    """ADVICE r4 / VERDICT r4 weak 1: a launch that runs into its wait bound is an ERROR on a
    default handle; only a handle that asked for it (auto_safe_mode=True) warns -- with the
    original error in the text --, switches once and repeats the call, and counts the switch.
    Host logic only: the wrapper of pygp_amd._lib around a stand-in method."""
    import threading
    import warnings
    from pygp_amd import _lib

    class Fake(object):
        def __init__(self, auto):
            self._lock = threading.RLock()
            self._auto_safe = auto
            self.safe_mode_switches = 0
            self._h = object()
            self.calls = 0
            self.switched = []
            outer = self

            class L(object):
                @staticmethod
                def gpx_set_safe_mode(h, on):
                    outer.switched.append(on)
                    return 0
            self._L = L

        def work(self):
            self.calls += 1
            if not self.switched:
                raise _lib.GpxError('internal: the panel kernel timed out waiting for a dependency')
            return 42

        def broken(self):
            raise _lib.GpxError('hipErrorOutOfMemory')
    Fake.work = _lib._serialised(Fake.work)
    Fake.broken = _lib._serialised(Fake.broken)

    f = Fake(auto=False)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        with pytest.raises(_lib.GpxError, match='timed out waiting'):
            f.work()
    assert f.calls == 1 and f.switched == [] and f.safe_mode_switches == 0

    f = Fake(auto=True)
    with pytest.warns(RuntimeWarning, match='timed out waiting.*safe mode'):
        assert f.work() == 42
    assert f.calls == 2 and f.switched == [1] and f.safe_mode_switches == 1
    with pytest.raises(_lib.GpxError, match='OutOfMemory'):       # other errors pass through
        f.broken()
    # a second time-out on the same handle is not retried again
    f.switched.clear()
    with pytest.raises(_lib.GpxError, match='timed out waiting'):
        f.work()
