"""CPU tests of the batched-theta sharding (pygp_amd/batch.py) with a real
torch.distributed group: world_size 2 and 3, gloo backend. The device
evaluation (pygp_amd.batch._device_loglik / _device_posterior) is replaced by the
oracle in the worker processes (there is no GPU here); what is under test is the partition, the padding of ragged blocks
and the single all-gather."""

import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_everything():
    from pygp_amd.batch import partition
    for B in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            spans = [partition(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, grad, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import recipes
    from oracle import gp_oracle as orc
    import pygp_amd
    from pygp_amd.batch import loglik_batch_sharded
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port,
                            rank=rank, world_size=world)
    D, N = 2, 40
    X, y, _ = recipes.synthetic(N, D)
    kern = pygp_amd.kernels.SE(1.0, np.ones(D))
    spec = orc.se_spec(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    calls = []

    def oracle_eval(kernel, X_, y_, block, grad_, handle=None):
        calls.append(len(block))
        res = [orc.exact_eval(spec, th, X_, y_, grad=grad_) for th in block]
        if grad_:
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        return np.array(res)

    import pygp_amd.batch as batch_mod
    batch_mod._device_loglik = oracle_eval           # this worker process only
    out = loglik_batch_sharded(kern, thetas, X, y, grad=grad)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out, calls))


@pytest.mark.parametrize('world,B,grad', [(2, 8, False), (2, 5, True), (3, 7, True)])
def test_sharded_batch_gloo(world, B, grad):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import recipes
    from oracle import gp_oracle as orc
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, grad, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    D, N = 2, 40
    X, y, _ = recipes.synthetic(N, D)
    spec = orc.se_spec(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    want = [orc.exact_eval(spec, th, X, y, grad=True) for th in thetas]
    want_lZ = np.array([w[0] for w in want])
    want_dlZ = np.array([w[1] for w in want])
    total_calls = 0
    for rank, out, calls in results:
        lZ, dlZ = out if grad else (out, None)
        np.testing.assert_allclose(lZ, want_lZ, rtol=1e-13)     # every rank: full vector
        if grad:
            np.testing.assert_allclose(dlZ, want_dlZ, rtol=1e-13)
        total_calls += sum(calls)
    assert total_calls == B                                      # no theta done twice


def _oracle_posteriors(spec, thetas, X, y, Xs, grad):
    from oracle import gp_oracle as orc
    rows = []
    for th in thetas:
        s = orc.spec_set_hyper(orc._deepcopy_spec(spec), th[1:-1])
        R, a = orc.exact_update(s, th[0], th[-1], X, y)
        if grad:
            rows.append(orc.exact_posterior_grad(s, th[-1], X, R, a, Xs))
        else:
            rows.append(orc.exact_posterior(s, th[-1], X, R, a, Xs))
    return tuple(np.array(p) for p in zip(*rows))


def _post_worker(rank, world, port, B, grad, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import recipes
    from oracle import gp_oracle as orc
    import pygp_amd
    from pygp_amd.batch import posterior_batch_sharded
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port,
                            rank=rank, world_size=world)
    D, N = 2, 30
    X, y, Xs = recipes.synthetic(N, D, n_test=5)
    kern = pygp_amd.kernels.SE(1.0, np.ones(D))
    spec = orc.se_spec(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    calls = []

    def oracle_eval(kernel, X_, y_, block, Xs_, grad_, handle=None):
        calls.append(len(block))
        return _oracle_posteriors(spec, block, X_, y_, Xs_, grad_)

    import pygp_amd.batch as batch_mod
    batch_mod._device_posterior = oracle_eval        # this worker process only
    out = posterior_batch_sharded(kern, thetas, X, y, Xs, grad=grad)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out, calls))


@pytest.mark.parametrize('world,B,grad', [(2, 5, False), (3, 4, True)])
def test_sharded_posterior_gloo(world, B, grad):
    """[m.posterior(X, grad) for m in samples] (mcmc.py:75-77) sharded over ranks:
    ragged blocks, one all-gather, every rank ends with all B models."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import recipes
    from oracle import gp_oracle as orc
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_post_worker, args=(r, world, port, B, grad, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    D, N = 2, 30
    X, y, Xs = recipes.synthetic(N, D, n_test=5)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    want = _oracle_posteriors(orc.se_spec(1.0, np.ones(D)), thetas, X, y, Xs, grad)
    total = 0
    for rank, out, calls in results:
        assert len(out) == len(want)
        for got, w in zip(out, want):
            assert got.shape == w.shape
            np.testing.assert_allclose(got, w, rtol=1e-13, atol=1e-15)
        total += sum(calls)
    assert total == B


def test_mixture_posterior_matches_the_reference_formulas():
    """mcmc.py:75-93 (uniform) and smc.py:128-150 (weighted), written out with the
    reference's own expressions."""
    from pygp_amd.batch import mixture_posterior
    rng = np.random.RandomState(3)
    B, m, d = 6, 4, 2
    mu_, s2_ = rng.randn(B, m), rng.rand(B, m)
    dmu_, ds2_ = rng.randn(B, m, d), rng.randn(B, m, d)
    mu = np.mean(mu_, axis=0)
    s2 = np.mean(s2_ + (mu_ - mu) ** 2, axis=0)
    dmu = np.mean(dmu_, axis=0)
    Dmu = dmu_ - dmu
    ds2 = np.mean(ds2_ + 2 * mu_[:, :, None] * Dmu - 2 * mu[None, :, None] * Dmu, axis=0)
    got = mixture_posterior((mu_, s2_, dmu_, ds2_))
    for g, w in zip(got, (mu, s2, dmu, ds2)):
        np.testing.assert_allclose(g, w, rtol=1e-14)
    got2 = mixture_posterior((mu_, s2_))
    np.testing.assert_allclose(got2[0], mu, rtol=1e-14)
    np.testing.assert_allclose(got2[1], s2, rtol=1e-14)
    w = rng.rand(B)
    w /= w.sum()
    wmu = np.average(mu_, weights=w, axis=0)
    ws2 = np.average(s2_ + (mu_ - wmu) ** 2, weights=w, axis=0)
    got3 = mixture_posterior((mu_, s2_), weights=w)
    np.testing.assert_allclose(got3[0], wmu, rtol=1e-14)
    np.testing.assert_allclose(got3[1], ws2, rtol=1e-14)
