"""
Ensembles of GPs that share one data set and differ only in their
hyperparameters -- the state the reference's meta-models keep as a Python list
of model copies and walk with per-sample loops:

    MCMC.posterior   [m.posterior(X, grad) for m in samples]     meta/mcmc.py:75-93
    SMC.add_data     [m.loglikelihood() for m in samples], weights  meta/smc.py:90-126
    SMC.posterior    weighted moment matching                     meta/smc.py:128-150

Here the ensemble is a (B, nhyper) array next to ONE model; every per-sample
loop is one call of pygp_amd.batch (gpx_loglik_batch / gpx_posterior_batch on
each rank's GPU, one all-gather), so the meta-models reach all GPUs of a node.
The proposal steps of the samplers (slice sampling, learning/sampling.py) stay
host orchestration outside this class: they hand new hyperparameters to
`set_hypers`.

(paths relative to /root/reference/pygp/)
"""

import itertools

import numpy as np
from scipy.special import logsumexp

from . import batch

__all__ = ['HyperEnsemble']

_UIDS = itertools.count(1)           # never reused, unlike id()


class HyperEnsemble(object):
    def __init__(self, model, hypers, logweights=None, group=None, handle=None, ndev=None):
        """model: a pygp_amd ExactGP (template: likelihood / kernel structure, data);
        hypers: (B, model.nhyper) rows in the model's layout [like | kernel | mean];
        logweights: normalised log weights (None: uniform, the MCMC case);
        ndev: deal the members to the first ndev GPUs of the node from this one process
        (gpx_loglik_batch_multi / gpx_posterior_batch_multi: a host thread and handle per
        device inside the library, one RCCL all-gather, no torch.distributed); None: this
        process's device, or the ranks of `group` when torch.distributed is initialised."""
        self._model = model.copy()
        self._hypers = np.array(hypers, dtype=float, ndmin=2)
        if self._hypers.shape[1] != self._model.nhyper:
            raise ValueError('hypers must have %d columns' % self._model.nhyper)
        n = len(self._hypers)
        self._logweights = (np.zeros(n) - np.log(n) if logweights is None
                            else np.array(logweights, dtype=float))
        if self._logweights.shape != (n,):
            raise ValueError('one log weight per member')
        self._group = group
        self._handle = handle
        self._ndev = None if ndev is None else int(ndev)
        if self._ndev is not None and self._ndev < 1:
            raise ValueError('ndev must be positive')
        self._loglikes = None
        self._data_serial = 0         # bumped when the shared data set changes
        self._uid = next(_UIDS)

    def __deepcopy__(self, memo):
        """A copy is another ensemble: its data may diverge from the original's, so it
        must not share the original's residency token."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, val in self.__dict__.items():
            setattr(new, key, val if key in ('_group', '_handle') else copy.deepcopy(val, memo))
        new._uid = next(_UIDS)
        return new

    __copy__ = lambda self: self.__deepcopy__({})

    def _multi_token(self):
        """Identifies this ensemble's current data set to the library's per-device
        handles: X, y go to the devices once per data set, later calls evaluate on the
        resident copy (gpx_*_batch_multi with X == NULL)."""
        return (self._uid, self._data_serial)

    # -- container protocol of the reference's meta-models --------------------
    def __len__(self):
        return len(self._hypers)

    def __iter__(self):
        for h in self._hypers:
            yield self._model.copy(h)

    def __getitem__(self, i):
        """Member i as a model of its own (a copy of the template at hypers[i]), what
        an entry of the reference's list of samples is (sampling.py:146)."""
        if isinstance(i, slice):
            return [self._model.copy(h) for h in self._hypers[i]]
        return self._model.copy(self._hypers[i])

    @property
    def hypers(self):
        return self._hypers

    @property
    def logweights(self):
        return self._logweights

    @property
    def ndata(self):
        return self._model.ndata

    @property
    def data(self):
        return self._model.data

    def set_hypers(self, hypers):
        """New members (after a proposal / MCMC move); cached likelihoods go."""
        hypers = np.array(hypers, dtype=float, ndmin=2)
        if hypers.shape != self._hypers.shape:
            raise ValueError('expected an array of shape %s' % (self._hypers.shape,))
        self._hypers = hypers
        self._loglikes = None

    # -- batched evaluations ---------------------------------------------------
    def _require_data(self):
        if self.ndata == 0:
            raise ValueError('no data')
        return self._model.data

    def loglikelihoods(self, grad=False):
        """[m.loglikelihood(grad) for m in samples] (smc.py:113-114,125-126)."""
        X, y = self._require_data()
        if self._ndev is not None:
            from . import _lib
            out = _lib.loglik_batch_multi(self._model._kernel._kspec(), self._hypers, X, y,
                                          grad=grad, ndev=self._ndev,
                                          token=self._multi_token())
        else:
            out = batch.loglik_batch_sharded(self._model._kernel, self._hypers, X, y,
                                             grad=grad, group=self._group,
                                             handle=self._handle)
        self._loglikes = np.array(out[0] if grad else out)
        return out

    def posterior(self, X, grad=False):
        """Moment-matched mixture of the members' posteriors: uniform weights are
        MCMC.posterior (mcmc.py:75-93), particle weights SMC.posterior
        (smc.py:128-150)."""
        Xd, y = self._require_data()
        X = self._model._kernel.transform(X)
        if self._ndev is not None:
            from . import _lib
            parts = _lib.posterior_batch_multi(self._model._kernel._kspec(), self._hypers, X,
                                               Xd, y, grad=grad, ndev=self._ndev,
                                               token=self._multi_token())
        else:
            parts = batch.posterior_batch_sharded(self._model._kernel, self._hypers, Xd, y, X,
                                                  grad=grad, group=self._group,
                                                  handle=self._handle)
        return batch.mixture_posterior(parts, weights=np.exp(self._logweights))

    # -- the weight bookkeeping of SMC (smc.py:90-126) ---------------------------
    def ess(self):
        """Effective sample size 1 / sum(w^2) (smc.py:94)."""
        return float(np.exp(-logsumexp(2 * self._logweights)))

    def resample(self, rng=None):
        """Multinomial resampling to uniform weights (smc.py:95-100)."""
        rng = np.random if rng is None else rng
        n = len(self)
        idx = rng.choice(n, n, p=np.exp(self._logweights))
        self._hypers = self._hypers[idx]
        self._logweights = np.zeros(n) - np.log(n)
        if self._loglikes is not None:
            self._loglikes = self._loglikes[idx]
        return idx

    def add_data(self, X, y):
        """Append observations and reweight the members by the likelihood ratio
        after / before (smc.py:102-116, Del Moral et al. 2006, Eqs. 30-31)."""
        if self.ndata > 0 and self._loglikes is None:
            self.loglikelihoods()
        before = np.zeros(len(self)) if self.ndata == 0 else self._loglikes
        X = self._model._kernel.transform(X)
        y = self._model._likelihood.transform(y)
        if self._model._X is None:
            self._model._X, self._model._y = X.copy(), y.copy()
        else:
            self._model._X = np.r_[self._model._X, X]
            self._model._y = np.r_[self._model._y, y]
        if hasattr(self._model, '_data_changed'):
            self._model._data_changed()
        self._data_serial += 1
        after = np.asarray(self.loglikelihoods())
        self._logweights = self._logweights + after - before
        self._logweights -= logsumexp(self._logweights)
