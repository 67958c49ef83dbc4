"""Kernels of the accelerated path."""

from ._base import Kernel, RealKernel
from .stationary import SE, Matern, Periodic, RQ
from ._combo import SumKernel, ProductKernel

__all__ = ['SE', 'Matern', 'Periodic', 'RQ', 'SumKernel', 'ProductKernel', 'Kernel', 'RealKernel']
