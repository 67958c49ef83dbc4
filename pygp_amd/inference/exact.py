"""
Exact GP regression on the MI355X.

Same model-facing interface as the reference (`GP` in
/root/reference/pygp/inference/_base.py:27-242 and `ExactGP` in
/root/reference/pygp/inference/exact.py:20-143): hyper layout
[like | kernel | mean], add_data / set_hyper trigger `_update`,
`loglikelihood(grad)`, `posterior(X)`, `copy`, `reset`, `from_gp`, `data`,
`ndata`, `nhyper`, `_params`. The numerical work (K + sn^2 I, Cholesky,
triangular solves, K^-1, trace terms, predictive mean/variance) is done by
libgpx.so; this class owns a device handle, keeps X and y resident in HBM after
add_data, and sends only hyperparameters down and a few doubles back up.
"""

import numpy as np

from ..utils.models import Parameterized
from ..likelihoods import Gaussian
from .. import _lib

__all__ = ['GP', 'ExactGP']


class GP(Parameterized):
    """State machine of a GP model: hypers, data, and when to refactorise."""

    def __init__(self, likelihood, kernel, mean):
        self._likelihood = likelihood
        self._kernel = kernel
        self._mean = float(mean)
        self._X = None
        self._y = None
        self.nhyper = likelihood.nhyper + kernel.nhyper + 1

    # -- bookkeeping ------------------------------------------------------
    def reset(self):
        """Forget all data."""
        self._X = None
        self._y = None

    def __repr__(self):
        def hang(prefix, text):
            return prefix + ('\n' + ' ' * len(prefix)).join(text.splitlines())
        inner = ',\n'.join([hang('likelihood=', repr(self._likelihood)),
                            hang('kernel=', repr(self._kernel)),
                            hang('mean=', str(self._mean))])
        return hang(type(self).__name__ + '(', inner + ')')

    def _params(self):
        out = [('like.' + p[0],) + tuple(p[1:]) for p in self._likelihood._params()]
        out += [('kern.' + p[0],) + tuple(p[1:]) for p in self._kernel._params()]
        return out + [('mean', 1, False)]

    def get_hyper(self):
        return np.r_[self._likelihood.get_hyper(), self._kernel.get_hyper(),
                     self._mean]

    def set_hyper(self, hyper):
        nl, nk = self._likelihood.nhyper, self._kernel.nhyper
        self._likelihood.set_hyper(hyper[:nl])
        self._kernel.set_hyper(hyper[nl:nl + nk])
        self._mean = hyper[-1]
        if self.ndata > 0:
            self._update()

    @property
    def ndata(self):
        return 0 if self._X is None else self._X.shape[0]

    @property
    def data(self):
        return (self._X, self._y)

    def add_data(self, X, y):
        X = self._kernel.transform(X)
        y = self._likelihood.transform(y)
        if self._X is None:
            self._X, self._y = X.copy(), y.copy()
            self._data_changed()
            self._update()
            return
        # same flow as the reference (_base.py:132-141): try the incremental
        # update, refactorise from scratch if it is not available
        try:
            self._updateinc(X, y)
            self._X = np.r_[self._X, X]
            self._y = np.r_[self._y, y]
        except NotImplementedError:
            self._X = np.r_[self._X, X]
            self._y = np.r_[self._y, y]
            self._data_changed()
            self._update()

    def _updateinc(self, X, y):
        raise NotImplementedError

    def posterior(self, X, grad=False):
        return self._marg_posterior(self._kernel.transform(X), grad)

    def sample(self, X, m=None, latent=True, rng=None):
        """Joint samples of the posterior at X (_base.py:143-178): an n-vector, or an
        (m, n) array when m is given; latent=False adds the observation noise. rng:
        None -> NumPy's global state, an int seed or a RandomState."""
        import scipy.linalg as sla
        X = self._kernel.transform(X)
        flatten = m is None
        m = 1 if flatten else m
        n = len(X)
        if rng is None:
            rng = np.random.mtrand._rand
        elif not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        mu, Sigma = self._full_posterior(X)
        Sigma = Sigma + 1e-10 * np.eye(n)
        f = mu[None] + np.dot(rng.normal(size=(m, n)), sla.cholesky(Sigma))
        if not latent:
            f = self._likelihood.sample(f.ravel(), rng).reshape(m, n)
        return f.ravel() if flatten else f

    def sample_fourier(self, N, rng=None):
        raise NotImplementedError(
            'Fourier-basis function samples are outside the accelerated path')

    def _full_posterior(self, X):
        raise NotImplementedError


class ExactGP(GP):
    """Exact inference; the likelihood must be Gaussian (exact.py:28-35)."""

    def __init__(self, likelihood, kernel, mean):
        if not isinstance(likelihood, Gaussian):
            raise ValueError('exact inference requires a Gaussian likelihood')
        super(ExactGP, self).__init__(likelihood, kernel, mean)
        self._dev_ = None          # _lib.Handle, created on first use
        self._resident = False     # X, y uploaded to this handle
        self._factored = False     # device holds R, a for the current hypers
        self._appends_in_place = 0  # add_data calls served by gpx_exact_append

    # -- device state -------------------------------------------------------
    def _dev(self):
        if self._dev_ is None:
            self._dev_ = _lib.Handle()
            self._resident = False
        return self._dev_

    def _data_changed(self):
        self._resident = False
        self._factored = False

    def __deepcopy__(self, memo):
        """copy.deepcopy (Parameterized.copy, models.py:47-55) must not share a
        device handle: the clone gets hypers + host data and re-uploads and
        refactorises lazily."""
        import copy
        clone = type(self).__new__(type(self))
        memo[id(self)] = clone
        for key, val in self.__dict__.items():
            if key not in ('_dev_', '_resident', '_factored', '_appends_in_place'):
                setattr(clone, key, copy.deepcopy(val, memo))
        clone._dev_, clone._resident, clone._factored = None, False, False
        clone._appends_in_place = 0
        return clone

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_dev_'], state['_resident'], state['_factored'] = None, False, False
        return state

    @classmethod
    def from_gp(cls, gp):
        new = cls(gp._likelihood.copy(), gp._kernel.copy(), gp._mean)
        if gp.ndata > 0:
            new.add_data(*gp.data)
        return new

    def reset(self):
        super(ExactGP, self).reset()
        self._data_changed()

    # -- the hot path -------------------------------------------------------
    def _update(self):
        """K + sn^2 I -> R -> a on the device (exact.py:50-55)."""
        # scipy.linalg.cholesky(check_finite=True) at exact.py:54 refuses a matrix
        # with NaN/inf entries; they can only come from the data or the hypers
        if not (np.all(np.isfinite(self.get_hyper())) and
                (self._resident or (np.all(np.isfinite(self._X)) and
                                    np.all(np.isfinite(self._y))))):
            self._factored = False
            raise ValueError('array must not contain infs or NaNs')
        dev = self._dev()
        if not self._resident:
            dev.set_data(self._X, self._y)
            self._resident = True
        self._factored = False
        dev.exact_update(self._kernel._kspec(),
                         self._likelihood.get_hyper()[0], self._mean)
        self._factored = True

    def _updateinc(self, X, y):
        """Extend R and a by the new observations in O(n^2) on the device
        (exact.py:57-62) when the current factor can be extended in place."""
        if not (self._factored and self._resident):
            raise NotImplementedError
        if X.shape[1] != self._X.shape[1]:
            raise ValueError('new inputs have the wrong dimension')
        if not (np.all(np.isfinite(X)) and np.all(np.isfinite(y))):
            raise ValueError('array must not contain infs or NaNs')
        try:
            extended = self._dev().exact_append(X, y)
        except Exception:
            # a failed extension (not positive definite, device error) has overwritten
            # part of the resident factor; the model stays on its old data like the
            # reference after a failed chol_update, and the next use re-uploads and
            # refactorises from the host copy
            self._resident = False
            self._factored = False
            raise
        if not extended:
            raise NotImplementedError
        self._appends_in_place += 1

    def _ensure(self):
        if self.ndata > 0 and not self._factored:
            self._update()

    def loglikelihood(self, grad=False):
        """log marginal likelihood (and d/d hyper) (exact.py:118-143)."""
        if self.ndata == 0:
            raise ValueError('no data')
        self._ensure()
        return self._dev().exact_loglik(self._kernel.nhyper, grad)

    def _marg_posterior(self, X, grad=False):
        """Predictive mean and variance (exact.py:81-97)."""
        if self._X is None:
            prior = (np.full(X.shape[0], self._mean), self._kernel.dget(X))
            # constant mean, stationary kernel: flat prior gradients (exact.py:99-101)
            return prior + (np.zeros_like(X), np.zeros_like(X)) if grad else prior
        self._ensure()
        if X.shape[1] != self._X.shape[1]:
            raise ValueError('test inputs have the wrong dimension')
        if grad:                       # exact.py:99-116
            return self._dev().exact_posterior_grad(X)
        return self._dev().exact_posterior(X)

    def _full_posterior(self, X):
        """Mean vector and full covariance (exact.py:64-79)."""
        if self._X is None:
            return np.full(X.shape[0], self._mean), self._kernel.get(X)
        self._ensure()
        if X.shape[1] != self._X.shape[1]:
            raise ValueError('test inputs have the wrong dimension')
        return self._dev().exact_posterior_full(X)

    # gp._R / gp._a as the reference exposes them (upper factor, R^-T (y-m))
    @property
    def _R(self):
        if self.ndata == 0:
            return None
        self._ensure()
        return self._dev().exact_get_factor(self.ndata, True)[0]

    @property
    def _a(self):
        if self.ndata == 0:
            return None
        self._ensure()
        return self._dev().exact_get_factor(self.ndata, False)[1]
