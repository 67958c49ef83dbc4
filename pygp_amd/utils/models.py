"""
Hyperparameter plumbing shared by kernels, likelihoods and GP models.

Mirrors the interface of /root/reference/pygp/utils/models.py (Parameterized
:21-55, printable :58-76, get_params :83-93): a model exposes `_params()` ->
[(name, size, islog)], `get_hyper()`, `set_hyper(h)`, and `copy(hyper=None)`.
pygp's learning code (optimize / sample) only talks to models through these.
"""

import abc
import copy as _copy

import numpy as np

__all__ = ['Parameterized', 'printable', 'get_params']


class Parameterized(abc.ABC):
    """Something described by a flat vector of (mostly log-space) hypers."""

    @abc.abstractmethod
    def _params(self):
        """List of (name, size, islog) describing the hyper vector layout."""

    @abc.abstractmethod
    def get_hyper(self):
        """Flat hyperparameter vector."""

    @abc.abstractmethod
    def set_hyper(self, hyper):
        """Assign the flat hyperparameter vector."""

    def copy(self, hyper=None):
        """Deep copy; optionally give the copy new hypers (models.py:47-55)."""
        clone = _copy.deepcopy(self)
        if hyper is not None:
            clone.set_hyper(hyper)
        return clone


def get_params(obj):
    """Yield (name, slice, islog) for every named block of obj's hypers."""
    start = 0
    for name, size, islog in obj._params():
        yield name, slice(start, start + size), islog
        start += size


def printable(cls):
    """Class decorator: repr as ClassName(name=value, ...) with log-space
    blocks shown exponentiated."""
    def __repr__(self):
        hyper = self.get_hyper()
        items = []
        for name, block, islog in get_params(self):
            value = hyper[block]
            value = value[0] if len(value) == 1 else value
            items.append('%s=%s' % (name, np.exp(value) if islog else value))
        return '%s(%s)' % (type(self).__name__, ', '.join(items))
    cls.__repr__ = __repr__
    return cls
