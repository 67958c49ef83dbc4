"""Kernels of the accelerated path."""

from ._base import Kernel, RealKernel
from .stationary import SE, Matern, Periodic
from ._combo import SumKernel

__all__ = ['SE', 'Matern', 'Periodic', 'SumKernel', 'Kernel', 'RealKernel']
