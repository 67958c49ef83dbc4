"""
Sum of kernels. Hyper vector = concatenation of the parts' vectors, gradients
chained in part order (/root/reference/pygp/kernels/_combo.py:55-120,
_real.py:76-106). The sum is evaluated in ONE pass on the device (each part is
an epilogue term of the same distance tile), not part by part.
"""

import itertools

import numpy as np

from ._base import RealKernel
from .. import _lib

__all__ = ['SumKernel', 'flatten']


def flatten(cls, *kernels):
    """Associativity: splice the parts of nested `cls` instances."""
    out = []
    for k in kernels:
        out.extend(k._parts if isinstance(k, cls) else [k])
    return out


class SumKernel(RealKernel):
    def __init__(self, *parts):
        ok = all(isinstance(p, RealKernel) for p in parts) and \
            all(p.ndim == parts[0].ndim for p in parts)
        if not ok:
            raise ValueError('cannot add mismatched kernels')
        self._parts = [p.copy() for p in flatten(SumKernel, *parts)]
        self.nhyper = sum(p.nhyper for p in self._parts)
        self.ndim = self._parts[0].ndim

    def __repr__(self):
        head = type(self).__name__ + '('
        body = (',\n').join(repr(p) for p in self._parts) + ')'
        return ('\n' + ' ' * len(head)).join((head + body).splitlines())

    def _params(self):
        out = []
        for i, p in enumerate(self._parts):
            out += [('part%d.%s' % (i, q[0]),) + tuple(q[1:]) for q in p._params()]
        return out

    def get_hyper(self):
        return np.hstack([p.get_hyper() for p in self._parts])

    def set_hyper(self, hyper):
        at = 0
        for p in self._parts:
            p.set_hyper(hyper[at:at + p.nhyper])
            at += p.nhyper

    def _kspec(self):
        return _lib.KSpecHolder(_lib.KIND_SUM, False, self.ndim,
                                parts=[p._kspec() for p in self._parts])

    def dget(self, X):
        return sum(p.dget(X) for p in self._parts)

    def dgrad(self, X):
        return itertools.chain.from_iterable(p.dgrad(X) for p in self._parts)
