"""
Learning on top of the accelerated ExactGP: type-II maximum likelihood and
hyperparameter sampling. Both are control flow over get_hyper / set_hyper /
loglikelihood of the model (/root/reference/pygp/learning/); every objective or
log-probability call is one device evaluation.
"""

from .optimization import optimize
from .sampling import sample

__all__ = ['optimize', 'sample']
