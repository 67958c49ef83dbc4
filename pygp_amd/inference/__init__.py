from .exact import GP, ExactGP
from .basic import BasicGP

__all__ = ['GP', 'ExactGP', 'BasicGP']
