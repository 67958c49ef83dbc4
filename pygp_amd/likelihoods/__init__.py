"""Gaussian observation noise, the only likelihood exact inference admits
(/root/reference/pygp/likelihoods/gaussian.py:23-49, _base.py:41-46)."""

import numpy as np

from ..utils.models import Parameterized, printable

__all__ = ['Gaussian', 'Likelihood']


class Likelihood(Parameterized):
    def transform(self, y):
        return np.array(y, ndmin=1, dtype=float)


@printable
class Gaussian(Likelihood):
    """y = f + N(0, sigma^2); one hyper, log sigma."""

    def __init__(self, sigma):
        self._logsigma = np.log(float(sigma))
        self.nhyper = 1

    def _params(self):
        return [('sigma', 1, True)]

    @property
    def s2(self):
        """Noise variance sigma^2 = exp(2 log sigma)."""
        return np.exp(self._logsigma * 2)

    def get_hyper(self):
        return np.r_[self._logsigma]

    def set_hyper(self, hyper):
        self._logsigma = hyper[0]

    def sample(self, f, rng=None):
        if rng is None:
            rng = np.random.mtrand._rand
        elif not isinstance(rng, np.random.RandomState):
            rng = np.random.RandomState(rng)
        return f + rng.normal(size=len(f), scale=np.exp(self._logsigma))
