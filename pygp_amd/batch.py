"""
Batched hyperparameter evaluation sharded over the GPUs of a node.

The reference's only data-parallel structure is its Python loops over
independent hyperparameter samples (SMC particles smc.py:102-126, MCMC samples
mcmc.py:75-77, sample(raw=False) sampling.py:146 under /root/reference/pygp/).
Here B (theta, shared dataset) evaluations are block-partitioned over the ranks
of a torch.distributed group -- one process per GPU -- each rank runs its block
on its own MI355X through gpx_loglik_batch, and ONE all-gather (RCCL over xGMI
when the backend is nccl) assembles the log-likelihood vector on every rank.
A single GP never spans GPUs.
"""

import numpy as np

__all__ = ['partition', 'loglik_batch_sharded']


def partition(B, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first B % world ranks get
    one extra element."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _local_eval(kernel, X, y, thetas, grad, handle):
    from . import _lib
    if handle is None:
        handle = _lib.default_handle()
    handle.set_data(X, y)
    return handle.loglik_batch(kernel._kspec(), thetas, grad=grad)


def loglik_batch_sharded(kernel, thetas, X, y, grad=False, group=None,
                         handle=None, evaluator=None):
    """Evaluate lZ (and dlZ) for every row of `thetas` ([log sn | kernel hypers
    | mean], the reference layout) on the data (X, y).

    Without an initialised torch.distributed this is a single-GPU batch. With
    one, every rank must call it with the same arguments; rank r computes its
    block and the results are all-gathered so that every rank returns the full
    arrays. `evaluator(kernel, X, y, thetas_block, grad)` replaces the device
    evaluation (tests inject the oracle to exercise the sharding logic on CPU
    ranks; the product never passes it).
    """
    thetas = np.ascontiguousarray(thetas, dtype=np.float64)
    B, nth = thetas.shape
    dist = None
    try:
        import torch.distributed as dist_
        if dist_.is_available() and dist_.is_initialized():
            dist = dist_
    except ImportError:
        pass
    world = dist.get_world_size(group) if dist else 1
    rank = dist.get_rank(group) if dist else 0
    lo, hi = partition(B, world, rank)

    def run(block):
        if evaluator is not None:
            return evaluator(kernel, X, y, block, grad)
        return _local_eval(kernel, X, y, block, grad, handle)

    if hi > lo:
        out = run(thetas[lo:hi])
        lZ_loc, dlZ_loc = out if grad else (out, None)
    else:
        lZ_loc, dlZ_loc = np.empty(0), np.empty((0, nth))
    if world == 1:
        return (lZ_loc, dlZ_loc) if grad else lZ_loc

    import torch
    backend = dist.get_backend(group)
    device = torch.device('cuda', torch.cuda.current_device()) \
        if backend == 'nccl' else torch.device('cpu')
    width = 1 + (nth if grad else 0)
    cap = -(-B // world)                           # ceil: equal-sized slots
    send = torch.full((cap, width), float('nan'), dtype=torch.float64)
    if hi > lo:
        send[:hi - lo, 0] = torch.from_numpy(np.asarray(lZ_loc))
        if grad:
            send[:hi - lo, 1:] = torch.from_numpy(np.asarray(dlZ_loc))
    send = send.to(device)
    slots = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(slots, send, group=group)                # the one collective
    recv = torch.stack(slots).cpu().numpy()
    lZ = np.empty(B)
    dlZ = np.empty((B, nth)) if grad else None
    for r in range(world):
        a, b = partition(B, world, r)
        lZ[a:b] = recv[r, :b - a, 0]
        if grad:
            dlZ[a:b] = recv[r, :b - a, 1:]
    return (lZ, dlZ) if grad else lZ
