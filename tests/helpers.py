"""Builders that turn a recipe descriptor (tests/recipes.py) into an oracle
spec or a pygp_amd kernel."""

import numpy as np

from oracle import gp_oracle as orc


def oracle_spec(desc):
    kind = desc[0]
    if kind == 'se':
        return orc.se_spec(*desc[1], **desc[2])
    if kind == 'matern':
        return orc.matern_spec(*desc[1], **desc[2])
    if kind == 'periodic':
        return orc.periodic_spec(*desc[1])
    if kind == 'rq':
        return orc.rq_spec(*desc[1], **desc[2])
    if kind == 'sum':
        return orc.sum_spec(*[oracle_spec(d) for d in desc[1]])
    if kind == 'product':
        return orc.product_spec(*[oracle_spec(d) for d in desc[1]])
    raise ValueError(kind)


def amd_kernel(desc):
    import pygp_amd.kernels as pk
    kind = desc[0]
    if kind == 'se':
        return pk.SE(*desc[1], **desc[2])
    if kind == 'matern':
        return pk.Matern(*desc[1], **desc[2])
    if kind == 'periodic':
        return pk.Periodic(*desc[1])
    if kind == 'rq':
        return pk.RQ(*desc[1], **desc[2])
    if kind == 'sum':
        parts = [amd_kernel(d) for d in desc[1]]
        k = parts[0]
        for p in parts[1:]:
            k = k + p
        return k
    if kind == 'product':
        parts = [amd_kernel(d) for d in desc[1]]
        k = parts[0]
        for p in parts[1:]:
            k = k * p
        return k
    raise ValueError(kind)


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0
