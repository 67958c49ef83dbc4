"""
Batched hyperparameter evaluation sharded over the GPUs of a node.

The reference's only data-parallel structure is its Python loops over
independent hyperparameter samples (SMC particles smc.py:102-126, MCMC samples
mcmc.py:75-77, sample(raw=False) sampling.py:146 under /root/reference/pygp/).
Here B (theta, shared dataset) evaluations are block-partitioned over the ranks
of a torch.distributed group -- one process per GPU -- each rank runs its block
on its own MI355X through gpx_loglik_batch / gpx_posterior_batch, and ONE
all-gather (RCCL over xGMI when the backend is nccl) assembles the per-model
results on every rank. A single GP never spans GPUs.

  loglik_batch_sharded     [m.loglikelihood(grad) for m in samples]
  posterior_batch_sharded  [m.posterior(X, grad) for m in samples]
  mixture_posterior        the moment matching MCMC.posterior / SMC.posterior do
                           on those per-model results (mcmc.py:75-93, smc.py:128-150)
"""

import numpy as np

__all__ = ['partition', 'loglik_batch_sharded', 'posterior_batch_sharded',
           'mixture_posterior']


def partition(B, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first B % world ranks get
    one extra element."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _group_info(group):
    dist = None
    try:
        import torch.distributed as dist_
        if dist_.is_available() and dist_.is_initialized():
            dist = dist_
    except ImportError:
        pass
    world = dist.get_world_size(group) if dist else 1
    rank = dist.get_rank(group) if dist else 0
    return dist, world, rank


def _all_gather_rows(dist, group, local, B, world):
    """local: (hi - lo, width) rows of this rank's block -> (B, width) on every
    rank, with ONE all-gather of equal-sized slots."""
    import torch
    backend = dist.get_backend(group)
    device = torch.device('cuda', torch.cuda.current_device()) \
        if backend == 'nccl' else torch.device('cpu')
    width = local.shape[1]
    cap = -(-B // world)                           # ceil: equal-sized slots
    send = torch.full((cap, width), float('nan'), dtype=torch.float64)
    if local.shape[0]:
        send[:local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local))
    send = send.to(device)
    slots = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(slots, send, group=group)                # the one collective
    recv = torch.stack(slots).cpu().numpy()
    out = np.empty((B, width))
    for r in range(world):
        a, b = partition(B, world, r)
        out[a:b] = recv[r, :b - a]
    return out


def _handle(handle, X, y):
    from . import _lib
    if handle is None:
        handle = _lib.default_handle()
    handle.set_data(X, y)
    return handle


def _device_loglik(kernel, X, y, thetas, grad, handle=None):
    """This rank's block on its GPU: gpx_loglik_batch on the resident data. (The CPU tests
    of the sharding logic replace this function with the oracle; nothing in the
    product does.)"""
    return _handle(handle, X, y).loglik_batch(kernel._kspec(), thetas, grad=grad)


def _device_posterior(kernel, X, y, thetas, Xs, grad, handle=None):
    """This rank's block on its GPU: gpx_posterior_batch (see _device_loglik)."""
    return _handle(handle, X, y).posterior_batch(kernel._kspec(), thetas, Xs, grad=grad)


def loglik_batch_sharded(kernel, thetas, X, y, grad=False, group=None, handle=None):
    """Evaluate lZ (and dlZ) for every row of `thetas` ([log sn | kernel hypers
    | mean], the reference layout) on the data (X, y).

    Without an initialised torch.distributed this is a single-GPU batch. With
    one, every rank must call it with the same arguments; rank r computes its
    block and the results are all-gathered so that every rank returns the full
    arrays.
    """
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    B, nth = thetas.shape
    dist, world, rank = _group_info(group)
    lo, hi = partition(B, world, rank)

    if hi > lo:
        out = _device_loglik(kernel, X, y, thetas[lo:hi], grad, handle)
        lZ_loc, dlZ_loc = out if grad else (out, None)
    else:
        lZ_loc, dlZ_loc = np.empty(0), np.empty((0, nth))
    if world == 1:
        return (lZ_loc, dlZ_loc) if grad else lZ_loc

    local = np.asarray(lZ_loc, dtype=float).reshape(-1, 1)
    if grad:
        local = np.hstack([local, np.asarray(dlZ_loc, dtype=float).reshape(-1, nth)])
    full = _all_gather_rows(dist, group, local, B, world)
    return (full[:, 0], full[:, 1:]) if grad else full[:, 0]


def posterior_batch_sharded(kernel, thetas, X, y, Xs, grad=False, group=None, handle=None):
    """[m.posterior(Xs, grad) for m in samples] (mcmc.py:75-77, smc.py:128-130)
    for models that share (X, y) and differ in their hyperparameters `thetas`:
    returns mu, s2 of shape (B, m) [and dmu, ds2 of shape (B, m, d)], the same
    on every rank."""
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    Xs = np.ascontiguousarray(Xs, dtype=np.float64)
    B = thetas.shape[0]
    m, d = Xs.shape
    dist, world, rank = _group_info(group)
    lo, hi = partition(B, world, rank)
    nparts = 4 if grad else 2
    if hi > lo:
        parts = _device_posterior(kernel, X, y, thetas[lo:hi], Xs, grad, handle)
        parts = [np.asarray(p, dtype=float) for p in parts]
    else:
        parts = [np.empty((0, m)), np.empty((0, m)), np.empty((0, m, d)),
                 np.empty((0, m, d))][:nparts]
    if world > 1:
        local = np.hstack([p.reshape(hi - lo, -1) for p in parts])
        full = _all_gather_rows(dist, group, local, B, world)
        widths = [m, m, m * d, m * d][:nparts]
        cuts = np.cumsum([0] + widths)
        parts = [full[:, cuts[i]:cuts[i + 1]] for i in range(nparts)]
        parts = parts[:2] + [p.reshape(B, m, d) for p in parts[2:]]
    return tuple(parts)


def mixture_posterior(parts, weights=None):
    """Moments of the mixture of the per-model posteriors `parts` = (mu_, s2_[,
    dmu_, ds2_]): uniform weights are MCMC.posterior (mcmc.py:75-93), normalised
    particle weights are SMC.posterior (smc.py:128-150)."""
    mu_, s2_ = np.asarray(parts[0]), np.asarray(parts[1])
    w = None if weights is None else np.asarray(weights, dtype=float)
    mu = np.average(mu_, weights=w, axis=0)
    s2 = np.average(s2_ + (mu_ - mu) ** 2, weights=w, axis=0)
    if len(parts) == 2:
        return mu, s2
    dmu_, ds2_ = np.asarray(parts[2]), np.asarray(parts[3])
    dmu = np.average(dmu_, weights=w, axis=0)
    Dmu = dmu_ - dmu
    ds2 = np.average(ds2_ + 2 * mu_[:, :, None] * Dmu - 2 * mu[None, :, None] * Dmu,
                     weights=w, axis=0)
    return mu, s2, dmu, ds2
