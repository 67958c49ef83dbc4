"""
Kernel interface of the drop-in (same names and argument meaning as
/root/reference/pygp/kernels/_base.py:22-64 and _real.py:28-39), evaluated on
the GPU through libgpx.so. A kernel object only holds hyperparameters; all
arithmetic happens in pygp_amd/csrc/kmat.hip.
"""

import numpy as np

from ..utils.models import Parameterized
from .. import _lib

__all__ = ['Kernel', 'RealKernel']


class Kernel(Parameterized):
    """k(x, x') with hyperparameters. Subclasses provide `_kspec()`."""

    def __call__(self, x1, x2):
        return self.get(np.asarray(x1)[None], np.asarray(x2)[None])[0]

    def _kspec(self):
        raise NotImplementedError

    def transform(self, X):
        raise NotImplementedError

    # device handle used for evaluation; a GP model installs its own
    _handle = None

    def _dev(self):
        return self._handle if self._handle is not None else _lib.default_handle()

    def __deepcopy__(self, memo):
        # a copy must not share (or pickle) a device handle: copies re-attach
        # lazily (checkpoint/resume row)
        import copy
        clone = type(self).__new__(type(self))
        memo[id(self)] = clone
        for key, val in self.__dict__.items():
            if key != '_handle':
                setattr(clone, key, copy.deepcopy(val, memo))
        return clone

    def get(self, X1, X2=None):
        """(n1, n2) matrix of covariances; X2=None means X2=X1."""
        X1 = self.transform(X1)
        X2 = None if X2 is None else self.transform(X2)
        self._check_dim(X1, X2)
        return self._dev().kernel_get(self._kspec(), X1, X2)

    def grad(self, X1, X2=None):
        """Iterator over d k(X1, X2) / d hyper_i, in get_hyper() order."""
        X1 = self.transform(X1)
        X2 = None if X2 is None else self.transform(X2)
        self._check_dim(X1, X2)
        G = self._dev().kernel_grad(self._kspec(), X1, X2)
        return iter(G)

    def _check_dim(self, X1, X2):
        for X in (X1, X2):
            if X is not None and X.shape[1] != self.ndim:
                raise ValueError('kernel has ndim=%d but inputs have %d columns'
                                 % (self.ndim, X.shape[1]))


class RealKernel(Kernel):
    """Kernel over real vectors; `+` builds a SumKernel (_real.py:32-33)."""

    def __add__(self, other):
        from ._combo import SumKernel, flatten
        return SumKernel(*flatten(SumKernel, self, other))

    def __mul__(self, other):
        from ._combo import ProductKernel, flatten
        return ProductKernel(*flatten(ProductKernel, self, other))

    def transform(self, X):
        return np.array(X, ndmin=2, dtype=float)

    def gradx(self, X1, X2=None):
        """d k(x1, x2) / d x1, an (n1, n2, d) array (_real.py:41-46)."""
        return self._grad_inputs(X1, X2, 1)

    def grady(self, X1, X2=None):
        """d k(x1, x2) / d x2, an (n1, n2, d) array (_real.py:48-54)."""
        return self._grad_inputs(X1, X2, 2)

    def _grad_inputs(self, X1, X2, wrt):
        X1 = self.transform(X1)
        X2 = None if X2 is None else self.transform(X2)
        self._check_dim(X1, X2)
        return self._dev().kernel_gradx(self._kspec(), X1, X2, wrt)

    def gradxy(self, X1, X2=None):
        # mixed second derivatives are not on the accelerated path
        raise NotImplementedError

    def sample_spectrum(self, N, rng=None):
        raise NotImplementedError
