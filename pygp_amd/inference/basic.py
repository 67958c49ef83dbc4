"""
BasicGP: ExactGP + Gaussian noise + an SE/Matern kernel behind the reference's
convenience constructor and parameter names sn / sf / ell / mu
(/root/reference/pygp/inference/basic.py:19-70).
"""

import numpy as np

from ..utils.models import printable
from ..likelihoods import Gaussian
from ..kernels import SE, Matern
from .exact import ExactGP

__all__ = ['BasicGP']

_KERNELS = {
    'se': lambda sf, ell, ndim: SE(sf, ell, ndim),
    'matern1': lambda sf, ell, ndim: Matern(sf, ell, 1, ndim),
    'matern3': lambda sf, ell, ndim: Matern(sf, ell, 3, ndim),
    'matern5': lambda sf, ell, ndim: Matern(sf, ell, 5, ndim),
}


@printable
class BasicGP(ExactGP):
    def __init__(self, sn, sf, ell, mu=0, ndim=None, kernel='se'):
        if kernel not in _KERNELS:
            raise ValueError('Unknown kernel type')
        super(BasicGP, self).__init__(Gaussian(sn), _KERNELS[kernel](sf, ell, ndim),
                                      mu)

    def _params(self):
        return [('sn', 1, True)] + self._kernel._params() + [('mu', 1, False)]

    @classmethod
    def from_gp(cls, gp):
        if not isinstance(gp._likelihood, Gaussian):
            raise ValueError('BasicGP instances must have Gaussian likelihood')
        if isinstance(gp._kernel, SE):
            name = 'se'
        elif isinstance(gp._kernel, Matern):
            name = 'matern%d' % gp._kernel._d
        else:
            raise ValueError('BasicGP instances must have a SE/Matern kernel')
        new = cls(np.sqrt(gp._likelihood.s2), np.exp(gp._kernel._logsf),
                  np.exp(gp._kernel._logell), gp._mean,
                  ndim=gp._kernel.ndim if gp._kernel._iso else None, kernel=name)
        if gp.ndata > 0:
            new.add_data(*gp.data)
        return new
