"""pygp_amd.meta.HyperEnsemble: the per-sample loops of the reference's meta-models
(meta/mcmc.py:75-93, meta/smc.py:90-150) as batched calls. The CPU tests inject
the oracle as evaluator (no device); the GPU test compares the batched device
path with a Python loop over model copies, which is what the reference does."""

import numpy as np
import numpy.testing as nt
import pytest
from scipy.special import logsumexp

import recipes
from oracle import gp_oracle as orc

import pygp_amd
from pygp_amd.likelihoods import Gaussian
from pygp_amd.meta import HyperEnsemble

D = 2


def _oracle_evaluators(spec0):
    def loglik(kernel, X, y, block, grad, handle=None):
        res = [orc.exact_eval(spec0, th, X, y, grad=grad) for th in block]
        if grad:
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        return np.array(res)

    def posterior(kernel, X, y, block, Xs, grad, handle=None):
        rows = []
        for th in block:
            s = orc.spec_set_hyper(orc._deepcopy_spec(spec0), th[1:-1])
            R, a = orc.exact_update(s, th[0], th[-1], X, y)
            rows.append(orc.exact_posterior_grad(s, th[-1], X, R, a, Xs) if grad
                        else orc.exact_posterior(s, th[-1], X, R, a, Xs))
        return tuple(np.array(p) for p in zip(*rows))
    return loglik, posterior


def _template(n):
    X, y, Xs = recipes.synthetic(n, D, n_test=6)
    gp = pygp_amd.ExactGP(Gaussian(0.1), pygp_amd.kernels.SE(1.0, np.ones(D)), 0.0)
    return gp, X, y, Xs


def test_ensemble_on_cpu_with_the_oracle(monkeypatch):
    gp, X, y, Xs = _template(30)
    B = 5
    hypers = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    spec = orc.se_spec(1.0, np.ones(D))
    # no device here: the two device evaluations of pygp_amd.batch become the oracle
    import pygp_amd.batch as batch_mod
    ll_eval, post_eval = _oracle_evaluators(spec)
    monkeypatch.setattr(batch_mod, '_device_loglik', ll_eval)
    monkeypatch.setattr(batch_mod, '_device_posterior', post_eval)
    ens = HyperEnsemble(gp, hypers)
    assert len(ens) == B and ens.ndata == 0
    with pytest.raises(ValueError):
        ens.posterior(Xs)
    # SMC-style sequential data: weights follow the likelihood ratios (smc.py:102-116)
    ens.add_data(X[:20], y[:20])
    ll20 = np.array([orc.exact_eval(spec, th, X[:20], y[:20], grad=False) for th in hypers])
    want = ll20 - np.log(B)
    want -= logsumexp(want)
    nt.assert_allclose(ens.logweights, want, rtol=1e-12)
    ens.add_data(X[20:], y[20:])
    ll30 = np.array([orc.exact_eval(spec, th, X, y, grad=False) for th in hypers])
    want = want + ll30 - ll20
    want -= logsumexp(want)
    nt.assert_allclose(ens.logweights, want, rtol=1e-11)
    assert ens.ndata == 30 and abs(np.exp(ens.logweights).sum() - 1) < 1e-12
    nt.assert_allclose(ens.ess(), 1.0 / np.sum(np.exp(ens.logweights) ** 2), rtol=1e-12)
    # weighted mixture posterior, written out as in smc.py:128-150
    parts = _oracle_evaluators(spec)[1](None, X, y, hypers, Xs, True)
    w = np.exp(ens.logweights)
    mu_, s2_, dmu_, ds2_ = parts
    mu = np.average(mu_, weights=w, axis=0)
    s2 = np.average(s2_ + (mu_ - mu) ** 2, weights=w, axis=0)
    dmu = np.average(dmu_, weights=w, axis=0)
    Dmu = dmu_ - dmu
    ds2 = np.average(ds2_ + 2 * mu_[:, :, None] * Dmu - 2 * mu[None, :, None] * Dmu,
                     weights=w, axis=0)
    got = ens.posterior(Xs, grad=True)
    for g, wv in zip(got, (mu, s2, dmu, ds2)):
        nt.assert_allclose(g, wv, rtol=1e-12, atol=1e-14)
    # resampling: uniform weights, members drawn from the old ones (smc.py:95-100)
    old = ens.hypers.copy()
    idx = ens.resample(np.random.RandomState(0))
    nt.assert_allclose(ens.logweights, -np.log(B))
    nt.assert_array_equal(ens.hypers, old[idx])
    # a proposal step hands back new members
    ens.set_hypers(old)
    nt.assert_allclose(ens.loglikelihoods(), ll30, rtol=1e-13)
    with pytest.raises(ValueError):
        ens.set_hypers(old[:2])
    with pytest.raises(ValueError):
        HyperEnsemble(gp, hypers[:, :-1])


@pytest.mark.gpu
def test_ensemble_against_the_oracle_mixture():
    """The reference's way: a list of model copies, one likelihood / posterior call
    each, then moment matching (mcmc.py:75-93). The ensemble's two batched device
    calls are checked against that loop run on the ORACLE, and the loop over the
    ensemble's own model copies must agree too."""
    gp, X, y, Xs = _template(400)
    gp.add_data(X, y)
    B = 6
    hypers = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    spec = orc.se_spec(1.0, np.ones(D))
    ens = HyperEnsemble(gp, hypers)
    ll = ens.loglikelihoods()
    want_ll = [orc.exact_eval(spec, th, X, y, grad=False) for th in hypers]
    nt.assert_allclose(ll, want_ll, rtol=1e-8)
    ll_g, dll = ens.loglikelihoods(grad=True)
    for b in range(B):
        wl, wd = orc.exact_eval(spec, hypers[b], X, y)
        nt.assert_allclose(ll_g[b], wl, rtol=1e-8)
        nt.assert_allclose(dll[b], wd, rtol=1e-7, atol=1e-7 * np.max(np.abs(wd)))
    mu_, s2_, dmu_, ds2_ = _oracle_evaluators(spec)[1](None, X, y, hypers, Xs, True)
    mu = np.mean(mu_, axis=0)                        # mcmc.py:79-81
    s2 = np.mean(s2_ + (mu_ - mu) ** 2, axis=0)
    dmu = np.mean(dmu_, axis=0)                      # mcmc.py:86-91
    Dmu = dmu_ - dmu
    ds2 = np.mean(ds2_ + 2 * mu_[:, :, None] * Dmu - 2 * mu[None, :, None] * Dmu, axis=0)
    got = ens.posterior(Xs, grad=True)
    for g_, w_ in zip(got, (mu, s2, dmu, ds2)):
        nt.assert_allclose(g_, w_, rtol=1e-6, atol=1e-6)
    # the in-library multi-device route (ndev=1 here: same partition and code path as N
    # GPUs minus the gather) returns the bits of the single-device batch
    ens1 = HyperEnsemble(gp, hypers, ndev=1)
    assert np.array_equal(ens1.loglikelihoods(), ll)
    for g_, w_ in zip(ens1.posterior(Xs, grad=True), got):
        assert np.array_equal(g_, w_)
    # SMC-style weights after more data (smc.py:102-116) against the oracle
    Xn, yn, _ = recipes.synthetic(40, D, seed=5)
    ens.add_data(Xn, yn)
    X2, y2 = np.r_[X, Xn], np.r_[y, yn]
    after = np.array([orc.exact_eval(spec, th, X2, y2, grad=False) for th in hypers])
    want = -np.log(B) + after - np.array(want_ll)
    want -= logsumexp(want)
    nt.assert_allclose(ens.logweights, want, rtol=1e-6, atol=1e-6)
    # the list-of-copies view the reference's meta-models iterate over
    models = list(ens)
    assert len(models) == B and models[0].ndata == 440
    nt.assert_allclose([m.loglikelihood() for m in models], after, rtol=1e-8)


@pytest.mark.gpu
def test_sample_returns_an_ensemble_that_matches_the_reference(g_small):
    """learning.sample(..., raw=False) (sampling.py:80-146) on the device: the chain
    of the reference for the same seed, and the returned HyperEnsemble -- the n
    models of `[gp.copy(h) for h in hypers]` as ONE batched call each -- against the
    reference's per-model likelihoods and the moment-matched posterior its MCMC
    meta-model computes from that list (mcmc.py:75-84)."""
    from test_host_logic import _Uniform
    from pygp_amd.learning import sample
    X, y = g_small['xy.X'], g_small['xy.y']
    gp = pygp_amd.BasicGP(sn=.1, sf=1, ell=.1, mu=0)
    gp.add_data(X, y)
    priors = dict((k, _Uniform(*v)) for k, v in recipes.SAMPLE_BOUNDS.items())
    ens = sample(gp, priors, recipes.SAMPLE_N, raw=False, rng=recipes.SAMPLE_SEED)
    assert isinstance(ens, HyperEnsemble) and len(ens) == recipes.SAMPLE_N
    nt.assert_allclose(ens.hypers, g_small['sample.hypers'], rtol=1e-7, atol=1e-7)
    nt.assert_allclose(gp.get_hyper(), g_small['sample.final_hyper'], rtol=1e-7, atol=1e-7)
    nt.assert_allclose(ens.loglikelihoods(), g_small['sample.loglikes'], rtol=1e-7)
    mu, s2 = ens.posterior(g_small['xy.grid'])
    nt.assert_allclose(mu, g_small['sample.mix_mu'], rtol=1e-6, atol=1e-6)
    nt.assert_allclose(s2, g_small['sample.mix_s2'], rtol=1e-6, atol=1e-6)
    # it still indexes / iterates as models
    m3 = ens[3]
    nt.assert_allclose(m3.get_hyper(), ens.hypers[3])
    nt.assert_allclose(m3.loglikelihood(), g_small['sample.loglikes'][3], rtol=1e-8)
    assert len(ens[:2]) == 2
