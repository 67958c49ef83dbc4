"""
pygp_amd -- MI355X-native exact-GP hot path behind pygp's Kernel / GP interface.

Only the path named in BASELINE.json is here: pairwise kernel evaluation
(SE / Matern / Periodic / RQ, sums of products) and ExactGP update /
log-likelihood (+gradient) / posterior, executed by hand-written HIP kernels in
libgpx.so; `batch` and `meta` route the per-sample loops of the
reference's meta-models through the batched entry points.
"""

from . import kernels
from . import likelihoods
from . import inference
from . import learning
from . import batch
from . import meta
from .inference import BasicGP, ExactGP
from .learning import optimize

__all__ = ['BasicGP', 'ExactGP', 'optimize', 'kernels', 'likelihoods',
           'inference', 'learning', 'batch', 'meta']
