"""
Sums and products of kernels. Hyper vector = concatenation of the parts'
vectors, gradients chained in part order
(/root/reference/pygp/kernels/_combo.py:55-146, _real.py:76-128).

A combination is evaluated in ONE pass on the device: each part is an epilogue
term of the same distance tile, multiplied into its product group and added
into the sum (pygp_amd/csrc/kmat.hip) - not part by part. The device form is a
sum of products of primitive kernels; any nesting of sums and products is
expanded into it (a product distributes over the sums among its factors, a
primitive the expansion repeats keeps one set of hyperparameters), up to
GPX_MAX_PARTS = 8 primitive factors in all.
"""

import itertools

import numpy as np

from ._base import RealKernel
from .. import _lib

__all__ = ['ComboKernel', 'SumKernel', 'ProductKernel', 'flatten']


def flatten(cls, *kernels):
    """Associativity: splice the parts of nested `cls` instances."""
    out = []
    for k in kernels:
        out.extend(k._parts if isinstance(k, cls) else [k])
    return out


def _all_but(values):
    """out[i] = product of every entry of `values` except the i-th."""
    out = []
    for i in range(len(values)):
        rest = [v for j, v in enumerate(values) if j != i]
        acc = np.ones_like(values[i])
        for v in rest:
            acc = acc * v
        out.append(acc)
    return out


class ComboKernel(RealKernel):
    _verb = 'combine'
    _kind = None

    def __init__(self, *parts):
        ok = all(isinstance(p, RealKernel) for p in parts) and \
            all(p.ndim == parts[0].ndim for p in parts)
        if not ok:
            raise ValueError('cannot %s mismatched kernels' % self._verb)
        self._parts = [p.copy() for p in parts]
        self.nhyper = sum(p.nhyper for p in self._parts)
        self.ndim = self._parts[0].ndim

    def __repr__(self):
        head = type(self).__name__ + '('
        body = (',\n').join(repr(p) for p in self._parts) + ')'
        return ('\n' + ' ' * len(head)).join((head + body).splitlines())

    def _leaves(self):
        out = []
        for p in self._parts:
            out.extend(p._leaves() if isinstance(p, ComboKernel) else [p])
        return out

    def _params(self):
        # flat numbering of the primitive kernels, also through nested
        # combinations (_combo.py:73-88)
        out = []
        for i, p in enumerate(self._leaves()):
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
        return _lib.KSpecHolder(self._kind, False, self.ndim,
                                parts=[p._kspec() for p in self._parts])


class SumKernel(ComboKernel):
    _verb = 'add'
    _kind = _lib.KIND_SUM

    def dget(self, X):
        return sum(p.dget(X) for p in self._parts)

    def dgrad(self, X):
        return itertools.chain.from_iterable(p.dgrad(X) for p in self._parts)


class ProductKernel(ComboKernel):
    _verb = 'multiply'
    _kind = _lib.KIND_PRODUCT

    def dget(self, X):
        out = np.ones(len(X))
        for p in self._parts:
            out = out * p.dget(X)
        return out

    def dgrad(self, X):
        rest = _all_but([p.dget(X) for p in self._parts])
        for r, p in zip(rest, self._parts):
            for g in p.dgrad(X):
                yield r * g
