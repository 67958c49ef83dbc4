"""
SE, Matern and Periodic kernels: hyperparameter containers whose get()/grad()
run on the MI355X (pygp_amd/csrc/kmat.hip).

Constructor semantics, hyper layout and error behaviour follow the reference:
  SE        /root/reference/pygp/kernels/se.py:25-51
  Matern    /root/reference/pygp/kernels/matern.py:25-67
  Periodic  /root/reference/pygp/kernels/periodic.py:31-50
  RQ        /root/reference/pygp/kernels/rq.py:22-52
"""

import numpy as np

from ._base import RealKernel
from ..utils.models import printable
from .. import _lib

__all__ = ['SE', 'Matern', 'Periodic', 'RQ']


class _ARDKernel(RealKernel):
    """Signal amplitude + per-dimension (ARD) or shared (iso) lengthscale."""

    def _init_ard(self, sf, ell, ndim):
        self._logsf = np.log(float(sf))
        self._logell = np.log(ell)
        self._iso = False
        self.ndim = np.size(self._logell)
        self.nhyper = 1 + np.size(self._logell)
        if ndim is not None:
            if np.size(self._logell) != 1:
                raise ValueError('ndim only usable with scalar lengthscales')
            self._logell = float(self._logell)
            self._iso = True
            self.ndim = ndim

    def _params(self):
        return [('sf', 1, True), ('ell', self.nhyper - 1, True)]

    def get_hyper(self):
        return np.r_[self._logsf, self._logell]

    def set_hyper(self, hyper):
        self._logsf = hyper[0]
        self._logell = hyper[1] if self._iso else hyper[1:]

    def dget(self, X):
        """k(x, x) = sf^2 for every stationary kernel here."""
        return np.exp(self._logsf * 2) * np.ones(len(X))

    def dgrad(self, X):
        yield 2 * self.dget(X)
        for _ in range(self.nhyper - 1):
            yield np.zeros(len(X))


@printable
class SE(_ARDKernel):
    """Squared exponential, k = sf^2 exp(-|x-x'|^2_ell / 2)."""

    def __init__(self, sf, ell, ndim=None):
        self._init_ard(sf, ell, ndim)

    def _kspec(self):
        return _lib.KSpecHolder(_lib.KIND_SE, self._iso, self.ndim,
                                self.get_hyper())


@printable
class Matern(_ARDKernel):
    """Matern with nu = d/2, d in {1, 3, 5}."""

    def __init__(self, sf, ell, d=3, ndim=None):
        self._init_ard(sf, ell, ndim)
        self._d = d
        if d not in (1, 3, 5):
            raise ValueError('d must be one of 1, 3, or 5')

    def _kspec(self):
        return _lib.KSpecHolder(_lib.KIND_MATERN[self._d], self._iso, self.ndim,
                                self.get_hyper())


@printable
class RQ(_ARDKernel):
    """Rational quadratic, k = sf^2 (1 + |x-x'|^2_ell / (2 alpha))^-alpha; hypers
    [log sf, log ell..., log alpha]."""

    def __init__(self, sf, ell, alpha, ndim=None):
        self._init_ard(sf, ell, ndim)
        self._logalpha = np.log(float(alpha))
        self.nhyper += 1

    def _params(self):
        return [('sf', 1, True), ('ell', self.nhyper - 2, True), ('alpha', 1, True)]

    def get_hyper(self):
        return np.r_[self._logsf, self._logell, self._logalpha]

    def set_hyper(self, hyper):
        self._logsf = hyper[0]
        self._logell = hyper[1] if self._iso else hyper[1:-1]
        self._logalpha = hyper[-1]

    def _kspec(self):
        return _lib.KSpecHolder(_lib.KIND_RQ, self._iso, self.ndim, self.get_hyper())


@printable
class Periodic(RealKernel):
    """k = sf^2 exp(-2 sin^2(pi |x-x'| / p) / ell^2), one input dimension."""

    def __init__(self, sf, ell, p):
        self._logsf = np.log(float(sf))
        self._logell = np.log(float(ell))
        self._logp = np.log(float(p))
        self.ndim = 1
        self.nhyper = 3

    def _params(self):
        return [('sf', 1, True), ('ell', 1, True), ('p', 1, True)]

    def get_hyper(self):
        return np.r_[self._logsf, self._logell, self._logp]

    def set_hyper(self, hyper):
        self._logsf, self._logell, self._logp = hyper[0], hyper[1], hyper[2]

    def _kspec(self):
        return _lib.KSpecHolder(_lib.KIND_PERIODIC, False, self.ndim,
                                self.get_hyper())

    def dget(self, X):
        return np.exp(self._logsf * 2) * np.ones(len(X))

    def dgrad(self, X):
        yield 2 * self.dget(X)
        yield np.zeros(len(X))
        yield np.zeros(len(X))
