"""
Type-II maximum likelihood on top of the accelerated ExactGP.

The reference's optimize() (/root/reference/pygp/learning/optimization.py:21-67)
is pure control flow over get_hyper / set_hyper / loglikelihood(True); it is
restated here only so that the drop-in can be exercised end to end
(tests/test_gpu_gp.py reproduces /root/reference/tests/test_learning.py). Each
objective call is one gpx_exact_update + one gpx_exact_loglik on the device.
"""

import numpy as np
import scipy.optimize as so

from ..utils.models import get_params

__all__ = ['optimize']


def optimize(gp, priors=None):
    """Maximise the marginal likelihood over the hypers of `gp` in place.
    priors: {name: None} freezes the named block (the only prior form the
    reference supports, optimization.py:47-52)."""
    start = gp.get_hyper()
    free = np.ones(gp.nhyper, dtype=bool)
    blocks = dict((name, block) for name, block, _ in get_params(gp))
    for name, prior in (priors or {}).items():
        if prior is not None:
            raise NotImplementedError('only fixing priors (None) are supported')
        free[blocks[name]] = False

    def negative_loglik(x):
        hyper = start.copy()
        hyper[free] = x
        gp.set_hyper(hyper)
        lZ, dlZ = gp.loglikelihood(True)
        return -lZ, -dlZ[free]

    x, _, _ = so.fmin_l_bfgs_b(negative_loglik, start[free])
    final = start.copy()
    final[free] = x
    gp.set_hyper(final)
