"""
Hyperparameter sampling for the accelerated ExactGP.

`sample(gp, priors, n, raw=True, rng=None)` has the interface and the chain of
the reference's slice sampler (/root/reference/pygp/learning/sampling.py:80-146:
one slice move per sample along a random direction through all active
hyperparameters, spearmint-style step-out and shrinkage, :24-74); the same seed
gives the same chain. Every log-probability is priors + ONE value-only device
evaluation (gp.set_hyper + gp.loglikelihood()).

What differs is the end: with raw=False the reference returns
`[gp.copy(h) for h in hypers]` (sampling.py:146) -- n deep copies, each
re-running its own update, which its meta-models then walk one by one
(meta/mcmc.py:75-77). Here raw=False returns a pygp_amd.meta.HyperEnsemble: the
same n models as ONE template + the (n, nhyper) array, whose
loglikelihoods() / posterior() are one batched device call each
(gpx_loglik_batch / gpx_posterior_batch, sharded over the GPUs of the node
when a process group is up). It still iterates / indexes as model copies.

Priors are duck-typed like in the reference ({name: object with
.logprior(values)}; None freezes the block); the reference's prior classes
(pygp/priors) are outside the accelerated path.
"""

import numpy as np

from ..utils.models import get_params

__all__ = ['sample']


def _rng(rng):
    if rng is None:
        return np.random.mtrand._rand
    if isinstance(rng, np.random.RandomState):
        return rng
    return np.random.RandomState(rng)


def _slice_move(logp, x, rng, width=1.0, max_expand=1000):
    """One slice-sampling move from x along a random unit direction."""
    u = rng.randn(x.shape[0])
    u = u / np.sqrt(np.sum(u ** 2))

    def along(t):
        return logp(x + t * u)

    hi = width * rng.rand()
    lo = hi - width
    level = np.log(rng.rand()) + along(0.0)
    grown = 0
    while along(lo) > level and grown < max_expand:      # step out, left then right
        grown += 1
        lo -= width
    grown = 0
    while along(hi) > level and grown < max_expand:
        grown += 1
        hi += width
    while True:                                           # shrink towards 0 until accepted
        t = (hi - lo) * rng.rand() + lo
        value = along(t)
        if np.isnan(value):
            raise RuntimeError('slice sampler: the log-probability is NaN')
        if value > level:
            return x + t * u
        if t < 0:
            lo = t
        elif t > 0:
            hi = t
        else:
            raise RuntimeError('slice sampler: the slice shrank to a point')


def sample(gp, priors, n, raw=True, rng=None):
    """Draw n hyperparameter samples of `gp` (in the model's layout, log-space where
    the model stores logs). priors: {parameter name: prior | None}; a prior is
    anything with .logprior(values), None holds the block at its current value.
    Returns the (n, nhyper) array, or with raw=False a HyperEnsemble of the n
    models. The model is left at the last sample, like the reference."""
    rng = _rng(rng)
    active = np.ones(gp.nhyper, dtype=bool)
    logged = np.zeros(gp.nhyper, dtype=bool)
    terms = []                                   # (block, prior) of the active priors
    for name, block, islog in get_params(gp):
        logged[block] = islog
        if name in priors:
            if priors[name] is None:
                active[block] = False
            else:
                terms.append((block, priors[name]))

    base = gp.get_hyper()
    base[logged] = np.exp(base[logged])          # priors live in the natural space

    def logprob(x):
        h = base.copy()
        h[active] = x
        total = 0.0
        for block, prior in terms:               # cheap first: -inf skips the device
            total += prior.logprior(h[block])
            if np.isinf(total):
                return total
        h[logged] = np.log(h[logged])
        gp.set_hyper(h)
        return total + gp.loglikelihood()

    hypers = np.tile(base, (n, 1))
    x = base[active].copy()
    for i in range(n):
        x = _slice_move(logprob, x, rng)
        hypers[i, active] = x
    hypers[:, logged] = np.log(hypers[:, logged])
    gp.set_hyper(hypers[-1])
    if raw:
        return hypers
    from ..meta import HyperEnsemble
    return HyperEnsemble(gp, hypers)
