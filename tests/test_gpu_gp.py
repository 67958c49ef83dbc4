"""GPU parity of the ExactGP path (update / loglikelihood / posterior) through
the drop-in classes and the C-ABI, against golden vectors generated from the
reference and against the oracle. Recipes from
/root/reference/tests/test_inference.py and tests/test_learning.py.

Tolerances (BASELINE.json north_star): <= 1e-8 relative on the log-likelihood,
<= 1e-6 on posterior mean / variance; gradients are held to 1e-7 relative
(scaled by the largest component) and, at the BASELINE configs, to 1e-8 per
component."""

import numpy as np
import numpy.testing as nt
import scipy.optimize as spop
import pytest

import recipes
from conftest import load_golden, run_child
from helpers import amd_kernel, oracle_spec
from oracle import gp_oracle as orc

import pygp_amd
from pygp_amd.likelihoods import Gaussian

pytestmark = pytest.mark.gpu

RTOL_LZ = 1e-8
TOL_POST = 1e-6


def assert_grad_close(got, want, rtol=1e-7):
    scale = np.max(np.abs(want))
    nt.assert_allclose(got, want, rtol=rtol, atol=rtol * scale)


def make_small(name):
    if name == 'exact':
        return pygp_amd.ExactGP(Gaussian(1), pygp_amd.kernels.SE(1, 1, ndim=2), 0.0)
    return pygp_amd.BasicGP(1, 1, 1, 0, ndim=2)


@pytest.mark.parametrize('name', ['exact', 'basic'])
def test_small_golden(g_small, name):
    g = lambda k: g_small['gp.%s.%s' % (name, k)]
    gp = make_small(name)
    X, y, Xs, ys = recipes.inference_points(2, 0.0)
    gp.add_data(X, y)
    nt.assert_allclose(gp.get_hyper(), g('hyper'))
    nt.assert_allclose(gp._R, g('R'), rtol=1e-12, atol=1e-14)
    nt.assert_allclose(gp._a, g('a'), rtol=1e-12)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, g('lZ'), rtol=RTOL_LZ)
    nt.assert_allclose(gp.loglikelihood(), g('lZ'), rtol=RTOL_LZ)
    assert_grad_close(dlZ, g('dlZ'), 1e-10)
    mu, s2 = gp.posterior(Xs)
    nt.assert_allclose(mu, g('mu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g('s2'), rtol=TOL_POST, atol=TOL_POST)
    mu_g, s2_g, dmu, ds2 = gp.posterior(Xs, grad=True)
    nt.assert_allclose(mu_g, mu, rtol=1e-12)
    nt.assert_allclose(s2_g, s2, rtol=1e-12)
    nt.assert_allclose(dmu, g('dmu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(ds2, g('ds2'), rtol=TOL_POST, atol=TOL_POST)
    # full posterior and joint samples (exact.py:64-79, test_inference.py:81-83): the
    # Cholesky of the 10 x 10 covariance amplifies differences, hence 1e-5
    fmu, fS = gp._full_posterior(gp._kernel.transform(Xs))
    nt.assert_allclose(fmu, g('full_mu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(fS, g('full_Sigma'), rtol=TOL_POST, atol=1e-9)
    nt.assert_allclose(fS, fS.T, rtol=0, atol=1e-12)
    nt.assert_allclose(fS.diagonal(), s2, rtol=1e-9, atol=1e-12)
    nt.assert_allclose(gp.sample(Xs, m=3, rng=5), g('sample_latent'), rtol=1e-5, atol=1e-5)
    nt.assert_allclose(gp.sample(Xs, m=2, latent=False, rng=6), g('sample_noisy'),
                       rtol=1e-5, atol=1e-5)
    nt.assert_allclose(gp.sample(Xs, rng=7), g('sample_flat'), rtol=1e-5, atol=1e-5)
    with pytest.raises(NotImplementedError):
        gp.sample_fourier(10)
    # test_inference.py:147-157 (hyper + 1, also after reset)
    gp2 = gp.copy(gp.get_hyper() + 1)
    nt.assert_allclose(gp2.loglikelihood(), g('lZ_p1'), rtol=RTOL_LZ)
    mu, s2 = gp2.posterior(Xs)
    nt.assert_allclose(mu, g('mu_p1'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g('s2_p1'), rtol=TOL_POST, atol=TOL_POST)
    # the original is untouched by the copy
    nt.assert_allclose(gp.loglikelihood(), g('lZ'), rtol=RTOL_LZ)
    # test_inference.py:37-41,132-145 (prior after reset; re-adding the data)
    gp3 = gp.copy()
    gp3.reset()
    mu, s2 = gp3.posterior(Xs)
    nt.assert_allclose(mu, g('mu_prior'))
    nt.assert_allclose(s2, g('s2_prior'))
    _, _, dmu0, ds20 = gp3.posterior(Xs, grad=True)          # test_inference.py:40
    assert dmu0.shape == Xs.shape and not dmu0.any() and not ds20.any()
    nt.assert_allclose(gp3.sample(Xs, m=2, rng=8), g('sample_prior'),   # :41
                       rtol=1e-6, atol=1e-6)
    gp3.set_hyper(gp3.get_hyper())
    gp3.add_data(*gp.data)
    mu, s2 = gp3.posterior(Xs)
    nt.assert_allclose(mu, g('mu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g('s2'), rtol=TOL_POST, atol=TOL_POST)


def test_add_data_twice_equals_batch():
    """test_inference.py:65-79: data added in two calls == added at once."""
    X, y, Xs, ys = recipes.inference_points(2, 0.0)
    gp1 = make_small('exact')
    gp1.add_data(X, y)
    gp1.add_data(Xs, ys)
    gp2 = make_small('exact')
    gp2.add_data(np.r_[X, Xs], np.r_[y, ys])
    assert gp1.ndata == 20
    nt.assert_allclose(gp1.posterior(Xs), gp2.posterior(Xs), rtol=1e-12)
    nt.assert_allclose(gp1.loglikelihood(), gp2.loglikelihood(), rtol=1e-12)
    from_ = pygp_amd.ExactGP.from_gp(gp1)
    nt.assert_allclose(from_.loglikelihood(), gp1.loglikelihood(), rtol=1e-12)
    basic = pygp_amd.BasicGP.from_gp(gp1)
    nt.assert_allclose(basic.loglikelihood(), gp1.loglikelihood(), rtol=1e-12)


def _oracle_model(spec, theta, X, y, Xs):
    """lZ, dlZ, mu, s2 of the oracle for the data (X, y)."""
    R, a = orc.exact_update(spec, theta[0], theta[-1], X, y)
    lZ, dlZ = orc.exact_loglik(spec, theta[0], X, R, a, True)
    mu, s2 = orc.exact_posterior(spec, theta[-1], X, R, a, Xs)
    return lZ, dlZ, mu, s2


def _assert_model_matches_oracle(gp, spec, X, y, Xs):
    want_lZ, want_dlZ, want_mu, want_s2 = _oracle_model(spec, gp.get_hyper(), X, y, Xs)
    nt.assert_allclose(gp.loglikelihood(), want_lZ, rtol=RTOL_LZ)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, want_lZ, rtol=RTOL_LZ)
    assert_grad_close(dlZ, want_dlZ)
    mu, s2 = gp.posterior(Xs)
    nt.assert_allclose(mu, want_mu, rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, want_s2, rtol=TOL_POST, atol=TOL_POST)


def test_incremental_append_against_the_oracle():
    """test_inference.py:65-79 (_updateinc == full _update) at a size with several
    128-blocks: points added one at a time, every step through gpx_exact_append in
    place -- also the step that opens a new 128-block (384 -> 385) -- and the model
    checked against the oracle on X[:i+1] (the reference pins chol_update the same
    way)."""
    D = 3
    X, y, Xs = recipes.synthetic(420, D, n_test=16)
    ell = [0.7, 0.9, 1.1]
    k = pygp_amd.kernels.Matern(0.9, ell, d=3)
    spec = orc.matern_spec(0.9, ell, d=3)
    gp = pygp_amd.ExactGP(Gaussian(0.2), k, 0.1)
    gp.add_data(X[:300], y[:300])
    for i in range(300, 420):
        gp.add_data(X[i:i + 1], y[i:i + 1])
        assert gp._factored and gp._resident
        if i in (300, 301, 383, 384, 385, 419):
            _assert_model_matches_oracle(gp, spec, X[:i + 1], y[:i + 1], Xs)
    assert gp.ndata == 420
    assert gp._appends_in_place == 120           # no step fell back to a refactorisation
    nt.assert_allclose(np.triu(gp._R), gp._R)
    R, a = orc.exact_update(spec, gp.get_hyper()[0], gp.get_hyper()[-1], X, y)
    nt.assert_allclose(gp._R, R, rtol=1e-9, atol=1e-11)
    nt.assert_allclose(gp._a, a, rtol=1e-9, atol=1e-11)
    # several points at once, spanning two block boundaries, also in place
    gp2 = pygp_amd.ExactGP(Gaussian(0.2), k.copy(), 0.1)
    gp2.add_data(X[:200], y[:200])
    gp2.add_data(X[200:420], y[200:420])
    assert gp2._appends_in_place == 1
    _assert_model_matches_oracle(gp2, spec, X, y, Xs)
    nt.assert_array_equal(gp2.data[0], gp.data[0])
    # hyperparameters change after appends: full refactorisation on the grown data
    gp2.set_hyper(gp2.get_hyper() + 0.05)
    _assert_model_matches_oracle(gp2, spec_with(spec, gp2), X, y, Xs)


def spec_with(spec, gp):
    s = orc._deepcopy_spec(spec)
    return orc.spec_set_hyper(s, gp._kernel.get_hyper())


def test_append_across_panel_blocks_and_capacity():
    """Appends on a factor whose diagonal blocks came from the panel kernel (n > 1024),
    in chunks that cross 128- and 1024-boundaries, then past the reserved capacity
    (the handle answers -3 and add_data refactorises, _base.py:132-141)."""
    D = 4
    N0, N1, N2 = 1000, 1300, 2300
    X, y, Xs = recipes.synthetic(N2, D, n_test=12)
    ell = np.linspace(0.6, 1.2, D)
    spec = orc.se_spec(1.1, ell)
    gp = pygp_amd.ExactGP(Gaussian(0.15), pygp_amd.kernels.SE(1.1, ell), -0.1)
    gp.add_data(X[:N0], y[:N0])
    gp.posterior(Xs)                                  # value-only factor: R^-1 incomplete
    steps = 0
    for lo in range(N0, N1, 50):
        gp.add_data(X[lo:lo + 50], y[lo:lo + 50])
        steps += 1
    assert gp._appends_in_place == steps
    _assert_model_matches_oracle(gp, spec, X[:N1], y[:N1], Xs)
    # capacity of the first upload: round_up(1000 + 256, 1024) = 2048 rows
    gp.add_data(X[N1:N2], y[N1:N2])
    assert gp._appends_in_place == steps and gp.ndata == N2
    _assert_model_matches_oracle(gp, spec, X, y, Xs)


def test_failed_append_leaves_a_consistent_model():
    """Duplicated points with (numerically) zero noise make the extended matrix
    singular: add_data raises like chol_update would, the model keeps its old data
    and works again (host and device stay in step)."""
    X, y, Xs = recipes.synthetic(60, 2, n_test=5)
    ell = [0.05, 0.05]
    gp = pygp_amd.BasicGP(1e-200, 1.0, ell)
    gp.add_data(X, y)
    lZ0 = gp.loglikelihood()
    with pytest.raises(np.linalg.LinAlgError):
        gp.add_data(X[:20], y[:20])
    assert gp.ndata == 60
    assert gp._R.shape == (60, 60)
    nt.assert_allclose(gp.loglikelihood(), lZ0, rtol=1e-12)
    gp.set_hyper(np.r_[np.log(0.1), 0.0, np.log(ell), 0.0])
    gp.add_data(X[:20], y[:20] + 0.01)               # fine with real noise
    assert gp.ndata == 80 and gp._appends_in_place == 1
    spec = orc.se_spec(1.0, ell)
    _assert_model_matches_oracle(gp, spec, np.r_[X, X[:20]], np.r_[y, y[:20] + 0.01], Xs)


def test_loglikelihood_gradient_fd():
    """test_inference.py:105-112."""
    gp = make_small('basic')
    X, y, _, _ = recipes.inference_points(2, 0.0)
    gp.add_data(X, y)
    x = gp.get_hyper()
    f = lambda h: gp.copy(h).loglikelihood()
    _, g1 = gp.loglikelihood(grad=True)
    g2 = spop.approx_fprime(x, f, 1e-8)
    nt.assert_allclose(g1, g2, rtol=1e-5, atol=1e-5)


def test_posterior_input_gradients_fd():
    """test_inference.py:159-169: d mu / d x and d s2 / d x against finite
    differences of posterior()."""
    gp = make_small('basic')
    X, y, Xs, _ = recipes.inference_points(2, 0.0)
    gp.add_data(X, y)
    _, _, G_mu, G_s2 = gp.posterior(Xs, grad=True)
    f_mu = lambda x: gp.posterior(x[None])[0][0]
    f_s2 = lambda x: gp.posterior(x[None])[1][0]
    F_mu = np.array([spop.approx_fprime(x, f_mu, 1e-8) for x in Xs])
    F_s2 = np.array([spop.approx_fprime(x, f_s2, 1e-8) for x in Xs])
    nt.assert_allclose(G_mu, F_mu, rtol=1e-6, atol=1e-6)
    nt.assert_allclose(G_s2, F_s2, rtol=1e-5, atol=1e-5)


def test_xy_demo_and_optimize(g_small):
    """demos/basic.py:14-24 + tests/test_learning.py:21-36."""
    X, y = g_small['xy.X'], g_small['xy.y']
    gp = pygp_amd.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, g_small['xy.lZ0'], rtol=RTOL_LZ)
    assert_grad_close(dlZ, g_small['xy.dlZ0'], 1e-10)
    mu, s2 = gp.posterior(g_small['xy.grid'])
    nt.assert_allclose(mu, g_small['xy.mu0'], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g_small['xy.s20'], rtol=TOL_POST, atol=TOL_POST)
    pygp_amd.optimize(gp)
    # L-BFGS trajectories agree to optimiser tolerance, not to rounding
    nt.assert_allclose(gp.loglikelihood(), g_small['xy.lZ_opt'], rtol=1e-6)
    nt.assert_allclose(gp.get_hyper(), g_small['xy.hyper_opt'], rtol=2e-3, atol=2e-3)
    mu, s2 = gp.posterior(g_small['xy.grid'])
    nt.assert_allclose(mu, g_small['xy.mu_opt'], rtol=1e-3, atol=1e-3)
    # the optimum itself reproduces exactly
    gp.set_hyper(g_small['xy.hyper_opt'])
    nt.assert_allclose(gp.loglikelihood(), g_small['xy.lZ_opt'], rtol=RTOL_LZ)
    gp = pygp_amd.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    pygp_amd.optimize(gp, {'sn': None})
    assert gp.get_hyper()[0] == np.log(0.1)                 # test_learning.py:36
    nt.assert_allclose(gp.get_hyper(), g_small['xy.hyper_opt_fixsn'], rtol=2e-3,
                       atol=2e-3)


@pytest.mark.parametrize('name', sorted(recipes.MID_CASES))
def test_mid_golden(g_mid, name):
    desc, D = recipes.MID_CASES[name]
    g = lambda key: g_mid['%s.%s' % (name, key)]
    X, y, Xs = recipes.synthetic(recipes.MID_N, D, n_test=32)
    gp = pygp_amd.ExactGP(Gaussian(0.1), amd_kernel(desc), 0.25)
    gp.add_data(X, y)
    nt.assert_allclose(gp.get_hyper(), g('hyper'), rtol=1e-15)
    R = gp._R
    nt.assert_allclose(R.diagonal(), g('Rdiag'), rtol=1e-9)
    nt.assert_allclose(R[::7, ::5], g('R_s'), rtol=1e-7, atol=1e-10)
    nt.assert_allclose(gp._a, g('a'), rtol=1e-7, atol=1e-9)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, g('lZ'), rtol=RTOL_LZ)
    assert_grad_close(dlZ, g('dlZ'))
    mu, s2 = gp.posterior(Xs)
    nt.assert_allclose(mu, g('mu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g('s2'), rtol=TOL_POST, atol=TOL_POST)
    _, _, dmu, ds2 = gp.posterior(Xs, grad=True)
    nt.assert_allclose(dmu, g('dmu'), rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(ds2, g('ds2'), rtol=TOL_POST, atol=TOL_POST)


def test_not_positive_definite_raises():
    """sla.cholesky raises LinAlgError at exact.py:54; duplicated points with
    (numerically) zero noise make K singular."""
    X = np.random.RandomState(0).rand(40, 2)
    X = np.r_[X, X]
    y = np.zeros(80)
    gp = pygp_amd.BasicGP(1e-200, 1.0, [1.0, 1.0])
    with pytest.raises(np.linalg.LinAlgError):
        gp.add_data(X, y)
    # and the model recovers with sane hypers
    gp.set_hyper(np.r_[np.log(0.1), 0.0, 0.0, 0.0, 0.0])
    assert np.isfinite(gp.loglikelihood())


def test_eval_and_batch_entry_points():
    """gpx_exact_eval (fused objective) and gpx_loglik_batch agree with
    update + loglik, and with the oracle."""
    from pygp_amd import _lib
    D, N = 3, 300
    X, y, _ = recipes.synthetic(N, D)
    k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(4)])
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    nt.assert_allclose(lZ, lZv, rtol=1e-14)
    spec = orc.se_spec(1.0, np.ones(D))
    for b in range(4):
        want_lZ, want_dlZ = orc.exact_eval(spec, thetas[b], X, y)
        nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ)
        assert_grad_close(dlZ[b], want_dlZ)
        kb = k.copy(thetas[b][1:-1])
        l1, d1 = dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b])   # deterministic
    t = dev.timings()
    assert set(t) >= {'potrf', 'trace_grad', 'kernel_build'}


def test_product_kernel_multi_tile():
    """Sum of products on several 128-tiles (padded N): objective, gradient and
    posterior with input gradients against the oracle."""
    N, D = 700, 3
    X, y, Xs = recipes.synthetic(N, D, n_test=33)
    desc = recipes.MID_CASES['sum_prod3'][0]
    gp = pygp_amd.ExactGP(Gaussian(0.2), amd_kernel(desc), 0.1)
    gp.add_data(X, y)
    spec = oracle_spec(desc)
    theta = gp.get_hyper()
    want_lZ, want_dlZ = orc.exact_eval(spec, theta, X, y)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, want_lZ, rtol=RTOL_LZ)
    assert_grad_close(dlZ, want_dlZ)
    R, a = orc.exact_update(spec, theta[0], theta[-1], X, y)
    mu, s2, dmu, ds2 = gp.posterior(Xs, grad=True)
    wmu, ws2, wdmu, wds2 = orc.exact_posterior_grad(spec, theta[-1], X, R, a, Xs)
    nt.assert_allclose(mu, wmu, rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, ws2, rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(dmu, wdmu, rtol=1e-6, atol=1e-6)
    nt.assert_allclose(ds2, wds2, rtol=1e-6, atol=1e-6)


def test_ragged_mid_size_against_oracle():
    """N = 3001 (24 tiles, the last one almost empty), Matern-3/2 ARD D = 5: the
    padded factorisation, gradient and both posterior paths (first call through the
    left-half inverses, later calls through the completed inverse) against the
    oracle computed on the spot."""
    N, D = 3001, 5
    X, y, Xs = recipes.synthetic(N, D, n_test=40)
    ell = np.linspace(0.6, 1.4, D)
    gp = pygp_amd.ExactGP(Gaussian(0.15), pygp_amd.kernels.Matern(1.3, ell, d=3), -0.2)
    gp.add_data(X, y)
    spec = orc.matern_spec(1.3, ell, d=3)
    theta = gp.get_hyper()
    R, a = orc.exact_update(spec, theta[0], theta[-1], X, y)
    want_mu, want_s2 = orc.exact_posterior(spec, theta[-1], X, R, a, Xs)
    _, _, want_dmu, want_ds2 = orc.exact_posterior_grad(spec, theta[-1], X, R, a, Xs)

    def check_grad():
        mu, s2, dmu, ds2 = gp.posterior(Xs, grad=True)
        nt.assert_allclose(mu, want_mu, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(s2, want_s2, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(dmu, want_dmu, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(ds2, want_ds2, rtol=TOL_POST, atol=TOL_POST)

    check_grad()                             # before R^-1 is completed (block solves)
    for _ in range(3):                       # call 1: recursive solve, 2+: one product
        mu, s2 = gp.posterior(Xs)
        nt.assert_allclose(mu, want_mu, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(s2, want_s2, rtol=TOL_POST, atol=TOL_POST)
    check_grad()                             # after: beta = W V with the whole inverse
    want_lZ, want_dlZ = orc.exact_loglik(spec, theta[0], X, R, a, True)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, want_lZ, rtol=RTOL_LZ)
    assert_grad_close(dlZ, want_dlZ)
    fmu, fS = gp._full_posterior(Xs)         # full covariance (exact.py:64-79)
    wmu, wS = orc.exact_full_posterior(spec, theta[-1], X, R, a, Xs)
    nt.assert_allclose(fmu, wmu, rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(fS, wS, rtol=TOL_POST, atol=TOL_POST)
    mu1, s21 = gp.posterior(Xs[:1])          # a single test point (split-k path)
    nt.assert_allclose(mu1, want_mu[:1], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s21, want_s2[:1], rtol=TOL_POST, atol=TOL_POST)


@pytest.mark.parametrize('N,sns', [(4096, (1e-2, 1e-3, 1e-4)), (8192, (1e-2, 1e-3))])
def test_small_noise_against_the_oracle(N, sns):
    """Ill-conditioned K + sn^2 I (SE on D = 2: K itself is numerically rank deficient, so
    cond(K + sn^2 I) ~ ||K|| / sn^2 = 1e7 .. 1e11) where the reference substitutes
    (LAPACK dpotrf + dtrtrs, exact.py:54-55,88) and this library multiplies by explicit
    inverses of diagonal blocks. Which path runs depends on N:
      N = 4096  add_data -> update as ONE panel launch over the whole matrix (row panels by
                substitution inside the launch, round 3), loglikelihood(True) completes
                R^-1 from the inverses of the 1024-blocks (trtri) and forms K^-1 (lauum);
      N = 8192  only the multi-block driver exists: update = right-looking sweep over eight
                1024-blocks with explicit-inverse row panels R[k,k+1:] = W_kk^T A[k,k+1:],
                then trtri + lauum; and the fused evaluation gpx_exact_eval = the sweep
                with R^-1 and K^-1 built block column by block column inside it.
    The objective, its gradient and the posterior mean differ from the oracle by no more
    than cond * eps -- the bound DESIGN.md section 4 states, itself the accuracy either
    side can claim. Measured (profiles/r04_cond_check_{4096,8192}.txt): three orders of
    magnitude below it."""
    from pygp_amd import _lib
    D = 2
    X, y, Xs = recipes.synthetic(N, D, n_test=50)
    ell = np.array([0.5, 0.7])
    spec = orc.se_spec(1.0, ell)
    K = orc.kernel_get(spec, X)
    eps = np.finfo(float).eps
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    for sn in sns:
        gp = pygp_amd.BasicGP(sn, 1.0, ell)
        gp.add_data(X, y)
        lZ, dlZ = gp.loglikelihood(True)
        mu, s2 = gp.posterior(Xs)
        th = gp.get_hyper()
        R, a = orc.exact_update(spec, th[0], th[-1], X, y)
        wl, wd = orc.exact_loglik(spec, th[0], X, R, a, True)
        wm, ws = orc.exact_posterior(spec, th[-1], X, R, a, Xs)
        cond = np.abs(K).sum(0).max() / sn ** 2 + 1          # >= cond_2(K + sn^2 I)
        assert abs(lZ - wl) <= cond * eps * abs(wl), (sn, lZ, wl)
        assert np.max(np.abs(dlZ - wd)) <= cond * eps * np.max(np.abs(wd)), sn
        assert np.max(np.abs(mu - wm)) <= cond * eps * np.max(np.abs(y)), sn
        assert np.max(np.abs(s2 - ws)) <= 1e-9, sn
        # the fused evaluation (inverse inside the sweep) on the same matrix
        kk = pygp_amd.kernels.SE(1.0, ell)
        l2, d2 = dev.exact_eval(kk._kspec(), th[0], th[-1], True)
        assert abs(l2 - wl) <= cond * eps * abs(wl), (sn, l2, wl)
        assert np.max(np.abs(d2 - wd)) <= cond * eps * np.max(np.abs(wd)), sn
    dev.close()


def test_two_handles_from_two_threads():
    """Distinct handles may be driven from distinct threads (ctypes drops the GIL
    during the calls): results equal the single-threaded ones bit for bit."""
    import threading
    from pygp_amd import _lib
    D = 3
    k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
    jobs = [(500, 11), (777, 12)]
    want, got = {}, {}

    def run(slot, N, seed, out):
        X, y, _ = recipes.synthetic(N, D, seed=seed)
        dev = _lib.Handle(0)
        dev.set_data(X, y)
        thetas = np.array([recipes.theta_sweep(D, b + seed) for b in range(5)])
        res = []
        for _ in range(3):
            res.append(dev.loglik_batch(k._kspec(), thetas, grad=True))
        out[slot] = res
        dev.close()

    for i, (N, seed) in enumerate(jobs):
        run(i, N, seed, want)
    threads = [threading.Thread(target=run, args=(i, N, seed, got))
               for i, (N, seed) in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(len(jobs)):
        for (l1, d1), (l2, d2) in zip(want[i], got[i]):
            assert np.array_equal(l1, l2) and np.array_equal(d1, d2)


def test_four_threads_random_work_on_one_gpu():
    """tools/soak_threads.py for 6 seconds: four host threads with a handle each, random
    sizes from one point to 2 600, single evaluations, batches in groups and batch posteriors
    side by side; every result bit-equal to the same call made alone. (Without the
    device-wide order of panel launches two of four threads ended in "the panel kernel timed
    out waiting for a dependency" within 30 s; with it 8 148 calls, all equal.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = run_child([sys.executable, os.path.join(root, 'tools', 'soak_threads.py'), '6'],
                    timeout=600)
    assert out.returncode == 0 and 'soak ok' in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


@pytest.mark.gpu
def test_four_threads_large_models_on_one_gpu():
    """The same with 1 000 ... 9 000 points for 8 seconds: single evaluations with the
    look-ahead streams, batches of one to nine thetas in groups -- side by side.
    (Until the end of round 4 a context decided at every call whether to use its look-ahead
    streams from a device-wide count of running batches; a batch started by ANOTHER thread
    between an update that had deferred its last K^-1 product and the gradient stage that
    joins it gave wrong gradients -- 2 of 2 010 calls, relative error up to 5e3. Now a
    property of the context: 2 521 calls in 60 s, all bit-equal.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = run_child([sys.executable, os.path.join(root, 'tools', 'soak_threads.py'), '8', 'big'],
                    timeout=600)
    assert out.returncode == 0 and 'soak ok' in out.stdout, (out.stdout[-800:], out.stderr[-3000:])


def test_safe_mode_against_the_oracle_and_the_automatic_switch():
    """gpx_set_safe_mode: diagonal blocks by recursion down to the 128-tile leaf, no task-queue
    launch (whose workgroups wait for each other) -- single evaluations, groups and posteriors
    against the oracle at sizes that normally take the leaf, one panel, a whole-matrix launch,
    the sweep and the multi-block driver (tools/check_safe_mode.py). And the switch itself,
    deterministically: with a 10-us wait bound (GPX_PANEL_TIMEOUT_US=10, a test
    hook) every task-queue launch ends in "timed out waiting"; a DEFAULT handle raises that
    error and stays as it is; a handle made with auto_safe_mode=True warns once (the warning
    carries the error), switches, repeats the call, reports safe_mode / safe_mode_switches /
    plan['safe_mode'], and everything it returns afterwards is within tolerance of the oracle.
    (tools/soak_safe_auto.py, which provokes real starvation with four threads, is a developer
    soak and no longer part of the suite.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, 'tools', 'check_safe_mode.py')
    out = run_child([sys.executable, tool], timeout=600)
    assert out.returncode == 0 and 'safe mode ok' in out.stdout, (out.stdout[-800:], out.stderr[-3000:])
    env = dict(os.environ, GPX_PANEL_TIMEOUT_US='10')
    out = run_child([sys.executable, tool, 'raise'], env=env, timeout=300)
    assert out.returncode == 0 and 'raise ok' in out.stdout, (out.stdout[-800:], out.stderr[-3000:])
    out = run_child([sys.executable, tool, 'auto'], env=env, timeout=600)
    assert out.returncode == 0 and 'auto ok' in out.stdout, (out.stdout[-800:], out.stderr[-3000:])


def test_posterior_batch_entry_point():
    """gpx_posterior_batch = [m.posterior(X, grad) for m in samples] (mcmc.py:75-77):
    every model against the oracle, and the mixture
    moments of MCMC.posterior / SMC.posterior through pygp_amd.batch."""
    from pygp_amd import _lib
    from pygp_amd.batch import posterior_batch_sharded, mixture_posterior
    D, N, B = 3, 333, 7
    X, y, Xs = recipes.synthetic(N, D, n_test=21)
    k = pygp_amd.kernels.Matern(1.0, np.linspace(.5, 1.5, D), d=3)
    spec = orc.matern_spec(1.0, np.ones(D), d=3)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    for grad in (False, True):
        parts = dev.posterior_batch(k._kspec(), thetas, Xs, grad=grad)
        assert len(parts) == (4 if grad else 2)
        for b in range(B):
            s = orc.spec_set_hyper(orc._deepcopy_spec(spec), thetas[b][1:-1])
            R, a = orc.exact_update(s, thetas[b][0], thetas[b][-1], X, y)
            want = orc.exact_posterior_grad(s, thetas[b][-1], X, R, a, Xs)
            for got, w in zip(parts, want):
                nt.assert_allclose(got[b], w, rtol=TOL_POST, atol=TOL_POST)
    # single-process call of the sharded front end + the mixture
    parts = posterior_batch_sharded(k, thetas, X, y, Xs, grad=True, handle=dev)
    mu, s2, dmu, ds2 = mixture_posterior(parts)
    nt.assert_allclose(mu, parts[0].mean(0), rtol=1e-13)
    assert s2.shape == (21,) and dmu.shape == (21, D) and ds2.shape == (21, D)
    assert np.all(s2 >= parts[1].mean(0) - 1e-15)
    # B = 0 and m = 0 edge cases
    e = dev.posterior_batch(k._kspec(), thetas[:0], Xs)
    assert e[0].shape == (0, 21)
    e = dev.posterior_batch(k._kspec(), thetas[:2], Xs[:0])
    assert e[0].shape == (2, 0)


def test_maunaloa_demo():
    """pygp/demos/maunaloa.py:27-41 on the device: SE + SE*Periodic + RQ + SE (five
    primitive kernels, n = 607) against the reference's own outputs."""
    g = load_golden('g_maunaloa.npz')
    X, y = g['X'], g['y']
    gp = pygp_amd.ExactGP(Gaussian(0.2), amd_kernel(recipes.MAUNALOA_KERNEL), y.mean())
    gp.add_data(X, y)
    nt.assert_allclose(gp.get_hyper(), g['hyper'], rtol=1e-15)
    lZ, dlZ = gp.loglikelihood(True)
    nt.assert_allclose(lZ, g['lZ'], rtol=RTOL_LZ)
    assert_grad_close(dlZ, g['dlZ'])
    mu, s2, dmu, ds2 = gp.posterior(g['Xs'], grad=True)
    nt.assert_allclose(mu, g['mu'], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g['s2'], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(dmu, g['dmu'], rtol=1e-6, atol=1e-6)
    nt.assert_allclose(ds2, g['ds2'], rtol=1e-6, atol=1e-6)


def _big(tag, idx=0):
    g = load_golden('g_%s.npz' % tag)
    cfg = recipes.BIG_CASES[tag]
    X, y, Xs = recipes.synthetic(cfg['N'], cfg['D'], n_test=16)
    theta = g['theta%d' % idx]
    gp = pygp_amd.ExactGP(Gaussian(1.0), amd_kernel(cfg['kernel']), 0.0)
    gp._X, gp._y = X, y
    gp._data_changed()
    gp.set_hyper(theta)
    lZ, dlZ = gp.loglikelihood(True)
    mu, s2 = gp.posterior(Xs)
    nt.assert_allclose(lZ, g['lZ%d' % idx], rtol=RTOL_LZ)
    assert_grad_close(dlZ, g['dlZ%d' % idx])
    # every component on its own, not only against the largest one (measured:
    # <= 1.1e-11, tools/grad_err.py)
    nt.assert_allclose(dlZ, g['dlZ%d' % idx], rtol=1e-8)
    nt.assert_allclose(mu, g['mu%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2, g['s2%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    a = gp._a
    nt.assert_allclose(a[:64], g['a_head%d' % idx], rtol=1e-7, atol=1e-9)
    # input gradients of the posterior at full size (exact.py:99-116; the reference pins
    # them at N = 10 by finite differences, tests/test_inference.py:159-169): here the
    # inverse factor comes from the panel kernel and the look-ahead driver's columns
    mu_g, s2_g, dmu, ds2 = gp.posterior(Xs, grad=True)
    nt.assert_allclose(mu_g, g['mu%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(s2_g, g['s2%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(dmu, g['dmu%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    nt.assert_allclose(ds2, g['ds2%d' % idx], rtol=TOL_POST, atol=TOL_POST)
    return gp


def test_config2_n4096():
    """BASELINE configs[1]: SE-ARD fp64 N=4096 D=8."""
    _big('c2')


def test_config4_n8192():
    """BASELINE configs[3]: first two thetas of the sweep, N=8192 D=8."""
    _big('c4', 0)
    _big('c4', 1)


def test_config4_batch_of_64():
    """BASELINE configs[3] as stated: 64 thetas x N=8192 D=8 through ONE
    gpx_loglik_batch call. Members 0 and 1 against the reference goldens, and every
    member bit-equal to its own single gpx_exact_eval, with gradients and value only (the
    members run in groups, pygp_amd/csrc/group.hip; the single evaluation on the look-ahead
    streams: same arithmetic, same bits)."""
    from pygp_amd import _lib
    g = load_golden('g_c4.npz')
    N, D, B = 8192, 8, 64
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    nt.assert_array_equal(thetas[0], g['theta0'])
    nt.assert_array_equal(thetas[1], g['theta1'])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    for b in (0, 1):
        nt.assert_allclose(lZ[b], g['lZ%d' % b], rtol=RTOL_LZ)
        assert_grad_close(dlZ[b], g['dlZ%d' % b])
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    nt.assert_allclose(lZv, lZ, rtol=1e-13)
    for b in range(B):
        kb = k.copy(thetas[b][1:-1])
        l1, d1 = dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b])
        assert dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], False) == lZv[b]
    dev.close()


def test_metric_config_n16384():
    """BASELINE metric config: SE-ARD fp64 N=16384 D=8."""
    _big('metric', 1)


def test_metric_config_group_of_six():
    """The exact call the headline times (bench.py: one step = gpx_loglik_batch over
    theta_eval(8, 0..5) at N = 16384, gradients on -- ONE group of six members in lock-step):
    member 0 against the reference's golden lZ1 / dlZ1 (theta_eval(8, 0) is the fixture's
    theta1; 1e-8 relative, every gradient component on its own), and every member bit-equal
    to the same theta through gpx_exact_eval (the look-ahead path that
    test_metric_config_n16384 pins). Matches /root/reference/pygp/inference/exact.py:118-143."""
    from pygp_amd import _lib
    g = load_golden('g_metric.npz')
    N, D = 16384, 8
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_eval(D, j) for j in range(6)])
    nt.assert_array_equal(thetas[0], g['theta1'])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    plan = dev.batch_plan(6, grad=True)
    assert plan['arrangement'] == 'groups/panel' and plan['members_per_group'] == 6 and \
        plan['groups_in_flight'] == 1 and plan['safe_mode'] is False, plan
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    nt.assert_allclose(lZ[0], g['lZ1'], rtol=RTOL_LZ)
    nt.assert_allclose(dlZ[0], g['dlZ1'], rtol=1e-8)
    for b in range(6):
        kb = k.copy(thetas[b][1:-1])
        l1, d1 = dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
    assert not dev.safe_mode and dev.safe_mode_switches == 0
    dev.close()


def test_config3_matern_n16384():
    """BASELINE configs[2]: Matern-5/2 ARD fp64 N=16384 D=16."""
    _big('c3', 0)


def test_full_size_properties():
    """Size-independent checks at the metric size that need no oracle:
    K^-1 K = I on probe vectors via the posterior identity (the posterior at the
    training inputs with tiny noise interpolates), determinism, and the
    mean-gradient identity sum(alpha) = 1^T K^-1 (y - m)."""
    N, D = 16384, 8
    X, y, _ = recipes.synthetic(N, D)
    gp = pygp_amd.BasicGP(0.1, 1.0, np.linspace(.5, 1.5, D))
    gp.add_data(X, y)
    lZ1, d1 = gp.loglikelihood(True)
    gp.set_hyper(gp.get_hyper())
    lZ2, d2 = gp.loglikelihood(True)
    assert lZ1 == lZ2 and np.array_equal(d1, d2)
    # d lZ / d mean by central differences on the value-only path
    h = gp.get_hyper()
    e = np.zeros_like(h)
    e[-1] = 1e-4
    fd = (gp.copy(h + e).loglikelihood() - gp.copy(h - e).loglikelihood()) / 2e-4
    nt.assert_allclose(d1[-1], fd, rtol=1e-6)
    # posterior at training points: mu = y - sn^2 alpha, 0 < s2 < sf^2
    idx = np.arange(0, N, 1024)
    mu, s2 = gp.posterior(X[idx])
    assert np.all(s2 > 0) and np.all(s2 < 1.0)
    assert np.max(np.abs(mu - y[idx])) < 0.5


def test_multi_device_entry_on_one_gpu():
    """gpx_loglik_batch_multi with ndev = 1 gives the bits of gpx_loglik_batch; with
    the RCCL gather forced on (a one-rank communicator: the only rehearsal of the
    collective a one-GPU box allows) the same bits come back through
    ncclAllGather; more devices than present is a clean error."""
    import json
    import os
    import sys
    from pygp_amd import _lib
    D, N, B = 3, 500, 7
    X, y, _ = recipes.synthetic(N, D)
    k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    thetas[3, 0] = -500.0                 # sn = e^-500 with duplicated points: not PD
    X[1] = X[0]
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    assert lZ[3] == -np.inf and np.all(np.isnan(dlZ[3])) and np.all(np.isfinite(lZ[:3]))
    mlZ, mdlZ = _lib.loglik_batch_multi(k._kspec(), thetas, X, y, grad=True, ndev=1)
    assert np.array_equal(mlZ, lZ) and np.array_equal(mdlZ, dlZ, equal_nan=True)
    again = _lib.loglik_batch_multi(k._kspec(), thetas[:2], grad=False, ndev=1)   # resident data
    assert np.array_equal(again, lZ[:2])
    assert _lib.loglik_batch_multi(k._kspec(), thetas[:0], X, y, ndev=1).shape == (0,)
    with pytest.raises(_lib.GpxError):
        _lib.loglik_batch_multi(k._kspec(), thetas, X, y, ndev=_lib.device_count() + 1)
    # the posterior entry: same partition and gather, the bits of gpx_posterior_batch
    Xs = recipes.synthetic(40, D, seed=5)[0]
    ref = dev.posterior_batch(k._kspec(), thetas, Xs, grad=True)
    got = _lib.posterior_batch_multi(k._kspec(), thetas, Xs, X, y, grad=True, ndev=1)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.all(np.isnan(got[0][3])) and np.all(np.isfinite(got[0][:3]))
    mu2, s22 = _lib.posterior_batch_multi(k._kspec(), thetas[:2], Xs, ndev=1)     # resident data
    assert np.array_equal(mu2, ref[0][:2]) and np.array_equal(s22, ref[1][:2])
    with pytest.raises(_lib.GpxError):
        _lib.posterior_batch_multi(k._kspec(), thetas, Xs, X, y, ndev=_lib.device_count() + 1)
    code = (
        "import sys, json, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "D, N, B = 3, 500, 7\n"
        "X, y, _ = recipes.synthetic(N, D)\n"
        "X[1] = X[0]\n"
        "k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))\n"
        "thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])\n"
        "thetas[3, 0] = -500.0\n"
        "lZ, dlZ = _lib.loglik_batch_multi(k._kspec(), thetas, X, y, grad=True, ndev=1)\n"
        "Xs = recipes.synthetic(40, D, seed=5)[0]\n"
        "mu, s2 = _lib.posterior_batch_multi(k._kspec(), thetas, Xs, ndev=1)\n"
        "print(json.dumps({'lZ': [repr(v) for v in lZ], 'dlZ': [repr(v) for v in dlZ.ravel()],\n"
        "                  'mu': [repr(v) for v in mu.ravel()], 's2': [repr(v) for v in s2.ravel()]}))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
         os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPX_MULTI_FORCE_RCCL='1')
    out = run_child([sys.executable, '-c', code], env=env,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert [repr(v) for v in lZ] == got['lZ']
    assert [repr(v) for v in dlZ.ravel()] == got['dlZ']
    assert [repr(v) for v in ref[0].ravel()] == got['mu']
    assert [repr(v) for v in ref[1].ravel()] == got['s2']


@pytest.mark.gpu
def test_cu_partition_switch_gives_the_same_bits():
    """The look-ahead driver has two arrangements: products on every CU with the two small
    products between diagonal blocks on the critical stream (default), and the round-2
    partition -- 32 CUs kept for the diagonal blocks, products on CU-masked streams,
    strict above np = 8192 (GPX_RESERVE_CUS=32). Both cut the same tile products, so one
    evaluation just above np = 8192 must give the same bits either way, and so must a
    small one. The child makes and closes two handles: the second takes the first one's
    masked streams over from the cache, and the teardown runs to the end."""
    import json
    import os
    import sys
    code = (
        "import sys, json, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "out = {}\n"
        "for N in (1500, 8300):\n"
        "    D = 4\n"
        "    X, y, _ = recipes.synthetic(N, D)\n"
        "    dev = _lib.Handle(0)\n"
        "    dev.set_data(X, y)\n"
        "    k = pygp_amd.kernels.SE(1.0, np.linspace(.5, 1.5, D))\n"
        "    th = recipes.theta_eval(D, 1)\n"
        "    lZ, dlZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)\n"
        "    out[str(N)] = [float(lZ).hex()] + [float(v).hex() for v in dlZ]\n"
        "print(json.dumps(out), flush=True)\n"
        "dev.close()    # normal teardown: CU-masked streams go back to the library's\n"
        "               # process-lifetime cache and are never destroyed (round 5; round 4\n"
        "               # skipped the teardown here because hipStreamDestroy of one could hang)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
         os.path.dirname(os.path.abspath(__file__)))
    got = []
    for reserve in ('0', '32'):
        env = dict(os.environ, GPX_RESERVE_CUS=reserve)
        out = run_child([sys.executable, '-c', code], env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        got.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1]
    assert all(np.isfinite(float.fromhex(v)) for v in got[0]['8300'])


def test_queue_probe_switch_gives_the_same_bits():
    """Where batches still run one context and stream per member (above np = 32768; everything
    with GPX_GROUP_MAX_NP=0) a context picks its stream by measuring how
    it runs beside the streams before it. Whatever it picks -- the probe on, off
    (pool order), full-mask streams with queues of their own instead of plain ones (round 3's
    pool) -- is an arrangement of streams, not of arithmetic: the same bits, and the bits of
    the groups."""
    import json
    import os
    import sys
    code = (
        "import sys, json, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "out = {}\n"
        "for N, B in ((8300, 3), (3000, 5)):\n"
        "    D = 4\n"
        "    X, y, _ = recipes.synthetic(N, D)\n"
        "    dev = _lib.Handle(0)\n"
        "    dev.set_data(X, y)\n"
        "    k = pygp_amd.kernels.SE(1.0, np.ones(D))\n"
        "    th = np.array([recipes.theta_sweep(D, b) for b in range(B)])\n"
        "    lZ, dlZ = dev.loglik_batch(k._kspec(), th, grad=True)\n"
        "    lZv = dev.loglik_batch(k._kspec(), th, grad=False)\n"
        "    out[str(N)] = [float(v).hex() for v in np.r_[lZ, dlZ.ravel(), lZv]]\n"
        "print(json.dumps(out), flush=True)\n"
        "dev.close()    # normal teardown: CU-masked streams go back to the library's\n"
        "               # process-lifetime cache and are never destroyed (round 5; round 4\n"
        "               # skipped the teardown here because hipStreamDestroy of one could hang)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
         os.path.dirname(os.path.abspath(__file__)))
    got = []
    for e in ({}, {'GPX_GROUP_MAX_NP': '0'}, {'GPX_GROUP_MAX_NP': '0', 'GPX_TWIN_PROBE': '0'},
              {'GPX_GROUP_MAX_NP': '0', 'GPX_TWIN_MASKED': '1'}):
        out = run_child([sys.executable, '-c', code], env=dict(os.environ, **e), timeout=600)
        assert out.returncode == 0, (e, out.stderr[-2000:])
        got.append(json.loads(out.stdout.strip().splitlines()[-1]))
    assert got[0] == got[1] == got[2] == got[3]


def test_multi_device_entry_with_faked_devices():
    """The in-library multi-device path has never met a node with more than one GPU. What
    can only fail at ndev > 1 -- a host thread and handle per device issuing launches side
    by side inside one process, the block partition, devices without members (B < ndev),
    NaN-padded slots, the scatter -- runs here with GPX_MULTI_FAKE=1: logical device i is
    physical device i % 1 and the concatenation stands in for the one ncclAllGather (RCCL
    refuses one GPU twice). Members come back bit-equal to the one-device batch, for
    likelihoods with gradients and for posteriors. (A child process: the switch is read
    once.)"""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "N, D, B = 1500, 3, 7\n"
        "X, y, Xs = recipes.synthetic(N, D, n_test=9)\n"
        "k = pygp_amd.kernels.SE(1.0, np.ones(D))\n"
        "thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])\n"
        "ref = _lib.loglik_batch_multi(k._kspec(), thetas, X, y, grad=True, ndev=1)\n"
        "refv = _lib.loglik_batch_multi(k._kspec(), thetas, grad=False, ndev=1)\n"
        "pref = _lib.posterior_batch_multi(k._kspec(), thetas, Xs, X, y, grad=True, ndev=1)\n"
        "for ndev in (2, 3, 8):\n"
        "    for b in (B, 2, 0):\n"
        "        got = _lib.loglik_batch_multi(k._kspec(), thetas[:b], X, y, grad=True, ndev=ndev)\n"
        "        assert np.array_equal(got[0], ref[0][:b]) and np.array_equal(got[1], ref[1][:b]), (ndev, b)\n"
        "        val = _lib.loglik_batch_multi(k._kspec(), thetas[:b], grad=False, ndev=ndev)\n"
        "        # (round 4: the order of arithmetic depends on (N, want_grad) only, whatever else\n"
        "        # the device is doing -- value-only members are bit-equal too)\n"
        "        assert np.array_equal(val, refv[:b]), (ndev, b)\n"
        "    pg = _lib.posterior_batch_multi(k._kspec(), thetas, Xs, X, y, grad=True, ndev=ndev)\n"
        "    assert all(np.array_equal(a, b_) for a, b_ in zip(pg, pref)), ndev\n"
        "print('faked devices ok')\n"
    ) % (root, os.path.join(root, 'tests'))
    env = dict(os.environ, GPX_MULTI_FAKE='1')
    out = run_child([sys.executable, '-c', code], env=env,
                         timeout=600)
    assert out.returncode == 0 and 'faked devices ok' in out.stdout, out.stderr[-3000:]


def test_large_batch_members_run_the_lookahead_and_keep_their_bits():
    """From np = 12288 (with gradients) the members of a batch run the multi-stream
    look-ahead of the single evaluation, each on streams of its own (round 3). Same
    arithmetic, different arrangement of streams: every member equals its own single
    gpx_exact_eval bit for bit, with gradients and value only."""
    from pygp_amd import _lib
    N, D, B = 12288, 8, 5
    X, y, _ = recipes.synthetic(N, D)
    thetas = np.array([recipes.theta_eval(D, 50 + b) for b in range(B)])
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    # (round 4: a batch above np = 8192 is one group in lock-step, pygp_amd/csrc/group.hip;
    # with GPX_GROUP_MIN_BIG=4 -- rounds 3-4 -- up to three members keep the contexts with
    # look-ahead: that arrangement runs in a child below. Both here.)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    nt.assert_allclose(lZv, lZ, rtol=1e-13)
    lZ3, dlZ3 = dev.loglik_batch(k._kspec(), thetas[:3], grad=True)
    assert np.array_equal(lZ3, lZ[:3]) and np.array_equal(dlZ3, dlZ[:3])
    assert np.array_equal(dev.loglik_batch(k._kspec(), thetas[:3], grad=False), lZv[:3])
    for b in range(B):
        kb = k.copy(thetas[b][1:-1])
        l1, d1 = dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b])
        assert dev.exact_eval(kb._kspec(), thetas[b][0], thetas[b][-1], False) == lZv[b]
    dev.close()
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json, numpy as np\n"
        "sys.path[:0] = [%r, %r]\n"
        "import recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "X, y, _ = recipes.synthetic(12288, 8)\n"
        "th = np.array([recipes.theta_eval(8, 50 + b) for b in range(3)])\n"
        "dev = _lib.Handle(0); dev.set_data(X, y)\n"
        "k = pygp_amd.kernels.SE(1.0, np.ones(8))\n"
        "assert dev.batch_plan(3, True)['arrangement'] == 'contexts', dev.batch_plan(3, True)\n"
        "lZ, dlZ = dev.loglik_batch(k._kspec(), th, grad=True)\n"
        "lZv = dev.loglik_batch(k._kspec(), th, grad=False)\n"
        "print(json.dumps([float(v).hex() for v in np.r_[lZ, dlZ.ravel(), lZv]]), flush=True)\n"
    ) % (root, os.path.join(root, 'tests'))
    out = run_child([sys.executable, '-c', code], env=dict(os.environ, GPX_GROUP_MIN_BIG='4'),
                    timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got == [float(v).hex() for v in np.r_[lZ[:3], dlZ[:3].ravel(), lZv[:3]]]


def test_first_block_does_not_outrun_the_rest_of_the_build():
    """The first diagonal block is factored on the rows of its own build while the rest
    of K is still being built (round 3). What follows it on the same stream -- at N >
    8192 the update of a 2048-block the lead rows do not cover -- has to wait for the whole
    matrix: found as a race under a busy batch, where the build is slow. The test hook
    holds the rest of the build back by 3 ms so that a missing wait shows every time;
    the evaluation must not notice."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "dev = _lib.Handle(0)\n"
        "for N in (4096, 12288):\n"
        "    X, y, _ = recipes.synthetic(N, 8)\n"
        "    dev.set_data(X, y)\n"
        "    th = recipes.theta_eval(8, 3)\n"
        "    k = pygp_amd.kernels.SE(1.0, np.ones(8)).copy(th[1:-1])\n"
        "    lZ, dlZ = dev.exact_eval(k._kspec(), th[0], th[-1], True)\n"
        "    lZv = dev.exact_eval(k._kspec(), th[0], th[-1], False)\n"
        "    print('RESULT', N, repr(float(lZ)), repr(float(lZv)), repr(float(np.abs(dlZ).sum())))\n"
    ) % (root, os.path.join(root, 'tests'))

    def run(env):
        out = run_child([sys.executable, '-c', code], env=env, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        return [l for l in out.stdout.splitlines() if l.startswith('RESULT')]

    plain = run(dict(os.environ))
    held = run(dict(os.environ, GPX_TEST_HOLD_BUILD_US='3000'))
    assert len(plain) == 2 and plain == held, (plain, held)
    assert all('inf' not in l and 'nan' not in l for l in plain), plain


def test_results_do_not_depend_on_launch_timing():
    """Race detection by perturbation (round 3): with GPX_TEST_JITTER a spinning one-wave
    kernel of pseudo-random length goes in front of every product, panel and build launch,
    on that launch's stream. Whatever is ordered by events stays ordered; whatever was only
    ordered by luck moves. Single evaluations on the multi-stream look-ahead (1024- and
    2048-blocks, with and without the inverse) and a batch whose members run the
    look-ahead side by side must give the same bits for every seed."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "dev = _lib.Handle(0)\n"
        "for N in (2000, 3001, 8192, 12288):\n"
        "    X, y, _ = recipes.synthetic(N, 8)\n"
        "    dev.set_data(X, y)\n"
        "    th = recipes.theta_eval(8, 5)\n"
        "    k = pygp_amd.kernels.SE(1.0, np.ones(8)).copy(th[1:-1])\n"
        "    lZ, dlZ = dev.exact_eval(k._kspec(), th[0], th[-1], True)\n"
        "    lZv = dev.exact_eval(k._kspec(), th[0], th[-1], False)\n"
        "    print('RESULT', N, float(lZ).hex(), float(lZv).hex(), ' '.join(float(v).hex() for v in dlZ))\n"
        "thetas = np.array([recipes.theta_eval(8, 60 + b) for b in range(4)])\n"
        "k = pygp_amd.kernels.SE(1.0, np.ones(8))\n"
        "lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)\n"
        "print('RESULT batch', ' '.join(float(v).hex() for v in lZ), ' '.join(float(v).hex() for v in dlZ.ravel()))\n"
    ) % (root, os.path.join(root, 'tests'))

    def run(env):
        out = run_child([sys.executable, '-c', code], env=env, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        return [l for l in out.stdout.splitlines() if l.startswith('RESULT')]

    plain = run(dict(os.environ))
    assert len(plain) == 5 and all('inf' not in l and 'nan' not in l for l in plain), plain
    for seed in ('1', '2:800', '3:100'):
        assert run(dict(os.environ, GPX_TEST_JITTER=seed)) == plain, seed


@pytest.mark.parametrize('D', [17, 24, 32])
def test_wide_inputs_against_oracle(D):
    """More than 16 input dimensions (up to GPX_MAX_DIM = 32) take the widest instances of
    the build, trace-gradient and posterior kernels, which no BASELINE config reaches: SE
    and Matern-5/2 with ARD lengthscales at N = 1100 (one panel launch plus a ragged
    tile), objective, all D + 3 gradient components and the posterior with its input
    gradients against the oracle computed on the spot."""
    N = 1100
    X, y, Xs = recipes.synthetic(N, D, n_test=30)
    ell = np.linspace(1.5, 3.0, D)
    for kern, spec in ((pygp_amd.kernels.SE(1.2, ell), orc.se_spec(1.2, ell)),
                       (pygp_amd.kernels.Matern(0.9, ell, d=5), orc.matern_spec(0.9, ell, d=5))):
        gp = pygp_amd.ExactGP(Gaussian(0.2), kern, 0.1)
        gp.add_data(X, y)
        theta = gp.get_hyper()
        R, a = orc.exact_update(spec, theta[0], theta[-1], X, y)
        want_lZ, want_dlZ = orc.exact_loglik(spec, theta[0], X, R, a, True)
        lZ, dlZ = gp.loglikelihood(True)
        assert dlZ.shape == (D + 3,)
        nt.assert_allclose(lZ, want_lZ, rtol=RTOL_LZ)
        assert_grad_close(dlZ, want_dlZ)
        wmu, ws2, wdmu, wds2 = orc.exact_posterior_grad(spec, theta[-1], X, R, a, Xs)
        mu, s2, dmu, ds2 = gp.posterior(Xs, grad=True)
        nt.assert_allclose(mu, wmu, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(s2, ws2, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(dmu, wdmu, rtol=TOL_POST, atol=TOL_POST)
        nt.assert_allclose(ds2, wds2, rtol=TOL_POST, atol=TOL_POST)


def test_whole_matrix_panel_launch_against_the_blocked_sweep():
    """Round 3: value-only factorisations up to np = 4096 run as ONE panel launch over all
    tiles of the matrix; it must leave behind what the blocked sweep leaves (R and the
    inverses of its 1024-blocks), so that the objective, the posterior (block substitution
    first, completed inverse later) and a gradient evaluation that follows agree with the
    blocked path to rounding (GPX_PANEL_WHOLE=0), also with the limit lowered so that the
    larger sizes take the blocked sweep and the smaller ones the launch. Ragged sizes
    included: a last 1024-block of one tile, and of seven."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, recipes, pygp_amd\n"
        "from pygp_amd import _lib\n"
        "dev = _lib.Handle(0)\n"
        "for N in (1100, 1930, 2560, 3001, 4096):\n"
        "    X, y, Xs = recipes.synthetic(N, 8, n_test=33)\n"
        "    dev.set_data(X, y)\n"
        "    th = recipes.theta_eval(8, 4)\n"
        "    k = pygp_amd.kernels.SE(1.0, np.ones(8)).copy(th[1:-1])\n"
        "    out = [dev.exact_eval(k._kspec(), th[0], th[-1], False)]\n"
        "    dev.exact_update(k._kspec(), th[0], th[-1])\n"
        "    out += list(dev.exact_posterior(Xs))            # block substitution\n"
        "    out += list(dev.exact_posterior_grad(Xs[:9]))   # completes the inverse\n"
        "    out += list(dev.exact_posterior(Xs))\n"
        "    lZ, dlZ = dev.exact_eval(k._kspec(), th[0], th[-1], True)\n"
        "    out += [lZ, dlZ]\n"
        "    np.save(sys.argv[1] + '_%%d.npy' %% N, np.concatenate([np.ravel(np.asarray(o, float)) for o in out]))\n"
    ) % (root, os.path.join(root, 'tests'))
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for tag, env in (('blocked', '0'), ('default', None), ('whole', '2048')):
            e = dict(os.environ)
            if env is not None:
                e['GPX_PANEL_WHOLE'] = env
            out = run_child([sys.executable, '-c', code, os.path.join(tmp, tag)], env=e,
                            timeout=600)
            assert out.returncode == 0, out.stderr[-3000:]
            res[tag] = {N: np.load(os.path.join(tmp, '%s_%d.npy' % (tag, N)))
                        for N in (1100, 1930, 2560, 3001, 4096)}
        for N, ref in res['blocked'].items():
            assert np.all(np.isfinite(ref))
            for tag in ('default', 'whole'):
                nt.assert_allclose(res[tag][N], ref, rtol=2e-9, atol=2e-9, err_msg='%s N=%d' % (tag, N))
        # the default takes the one-launch path at all of these sizes, the lowered limit
        # only up to it
        assert not np.array_equal(res['default'][4096], res['blocked'][4096])
        assert not np.array_equal(res['whole'][1930], res['blocked'][1930])
        assert np.array_equal(res['whole'][4096], res['blocked'][4096])


def test_appends_after_a_whole_matrix_launch():
    """The one-launch factorisation keeps its right-hand side in the padding columns right
    of the matrix (column np of A and of the staging matrix) -- where appended
    observations open their next 128-block. An update at N = 2040 (one launch, with the
    forward substitution in it), then 200 observations appended 40 at a time across the
    2048 boundary: the posterior equals that of a fresh update on all 2240 points."""
    from pygp_amd import _lib
    N, D = 2040, 8
    X, y, Xs = recipes.synthetic(N + 200, D, n_test=32)
    th = recipes.theta_eval(D, 9)
    kk = pygp_amd.kernels.SE(1.0, np.ones(D)).copy(th[1:-1])
    dev, ref = _lib.Handle(0), _lib.Handle(0)
    dev.set_data(X[:N], y[:N])
    dev.exact_update(kk._kspec(), th[0], th[-1])
    for i in range(0, 200, 40):
        dev.exact_append(X[N + i:N + i + 40], y[N + i:N + i + 40])
    mu, s2 = dev.exact_posterior(Xs)
    ref.set_data(X, y)
    ref.exact_update(kk._kspec(), th[0], th[-1])
    mu2, s22 = ref.exact_posterior(Xs)
    nt.assert_allclose(mu, mu2, rtol=1e-10, atol=1e-10)
    nt.assert_allclose(s2, s22, rtol=1e-10, atol=1e-10)
    dev.close()
    ref.close()
