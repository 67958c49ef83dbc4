"""Member-batched evaluation of batches (pygp_amd/csrc/group.hip; round 4): the thetas of a
gpx_loglik_batch / gpx_posterior_batch call advance in groups, every kernel one launch
over the whole group -- what replaces the per-sample loops of
/root/reference/pygp/meta/smc.py:102-126, /root/reference/pygp/meta/mcmc.py:75-77 and
/root/reference/pygp/learning/sampling.py:102-124.

What is pinned here: every member against the oracle, every member bit-equal to the
same evaluation on its own (gpx_exact_eval; the reference is deterministic per theta,
/root/reference/pygp/inference/exact.py:118-125), and results independent of the group
size, of the member's slot and of which of the two group arrangements -- one panel
launch with the members' task graphs interleaved, or the lock-step sweep -- runs."""

import os
import sys

import numpy as np
import numpy.testing as nt
import pytest

from conftest import run_child

import recipes
from helpers import amd_kernel, oracle_spec
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL_LZ = 1e-8                      # BASELINE.json north_star: <= 1e-8 rel on log-lik


def _thetas(desc, D, B, seed=0, spread=0.1):
    """B hyperparameter vectors [log sn | kernel | mean] around the descriptor's own."""
    k = amd_kernel(desc)
    base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
    rng = np.random.RandomState(4000 + seed)
    return k, base + spread * rng.randn(B, base.size)


def _single(dev, k, theta, grad):
    kb = k.copy(theta[1:-1])
    return dev.exact_eval(kb._kspec(), theta[0], theta[-1], grad)


@pytest.mark.parametrize('N,D,B', [(300, 3, 9), (1000, 8, 21), (1100, 2, 5)])
def test_group_members_against_oracle_and_single_evaluations(N, D, B):
    """SE-ARD members at one tile row (N = 300 pads to 384), one 1024-block and a whole-matrix
    launch with the right-hand side riding along (N = 1100): every member within the stated
    tolerance of the oracle and bit-equal to its own single evaluation, value-only and with
    gradients."""
    from pygp_amd import _lib
    X, y, _ = recipes.synthetic(N, D)
    desc = ('se', (1.0, list(np.linspace(0.5, 1.5, D))), {})
    k, thetas = _thetas(desc, D, B)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    spec = oracle_spec(desc)
    for b in range(B):
        want_lZ, want_dlZ = orc.exact_eval(spec, thetas[b], X, y)
        nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ)
        nt.assert_allclose(lZv[b], want_lZ, rtol=RTOL_LZ)
        assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * np.max(np.abs(want_dlZ))
        l1, d1 = _single(dev, k, thetas[b], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
        assert _single(dev, k, thetas[b], False) == lZv[b], b
    dev.close()


@pytest.mark.parametrize('name', ['matern5_ard16', 'sum_se_per1', 'sum_prod3', 'rq_ard8'])
def test_group_members_other_kernel_families(name):
    """Matern ARD (the row trace kernel), a sum with a periodic part, a sum of products and
    RQ (the generic trace kernel): member-batched kernel build and trace terms with more than
    one part per member."""
    from pygp_amd import _lib
    desc, D = recipes.MID_CASES[name]
    N, B = 700, 6
    X, y, _ = recipes.synthetic(N, D)
    k, thetas = _thetas(desc, D, B, seed=1, spread=0.05)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    for b in range(B):
        spec = orc.spec_set_hyper(oracle_spec(desc), thetas[b][1:-1])
        R, a = orc.exact_update(spec, thetas[b][0], thetas[b][-1], X, y)
        want_lZ, want_dlZ = orc.exact_loglik(spec, thetas[b][0], X, R, a, True)
        nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ)
        assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * np.max(np.abs(want_dlZ))
        l1, d1 = _single(dev, k, thetas[b], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
        assert _single(dev, k, thetas[b], False) == lZv[b], b
    dev.close()


def test_a_member_that_is_not_positive_definite():
    """Duplicated points and (numerically) no noise: that member comes back as -inf / NaN
    with its pivot in info (scipy's LinAlgError at exact.py:54 for a single model), the
    members around it are untouched."""
    from pygp_amd import _lib
    X = np.random.RandomState(0).rand(150, 2)
    X = np.r_[X, X]
    y = np.sin(X.sum(1))
    k = amd_kernel(('se', (1.0, [1.0, 1.0]), {}))
    good = np.r_[np.log(0.1), 0.0, 0.0, 0.0, 0.0]
    thetas = np.array([good, good + 0.01, np.r_[-460.0, 0.0, 0.0, 0.0, 0.0], good - 0.01])
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    info = np.zeros(4, dtype=np.int32)
    lZ = np.empty(4)
    dlZ = np.empty((4, 5))
    spec = k._kspec()
    _lib.check(dev._L.gpx_loglik_batch(dev._h, spec.ref(), _lib._ptr(thetas), 4, 1,
                                       _lib._ptr(lZ), _lib._ptr(dlZ), _lib._ptr(info)))
    assert lZ[2] == -np.inf and np.all(np.isnan(dlZ[2])) and info[2] > 0
    assert np.all(info[[0, 1, 3]] == 0)
    for b in (0, 1, 3):
        l1, d1 = _single(dev, k, thetas[b], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
    dev.close()


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import recipes, pygp_amd
from pygp_amd import _lib
out = {}
for N, D, B in %(cases)r:
    X, y, _ = recipes.synthetic(N, D)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    # the members in another order, and a shorter batch: other slots, other groups
    perm = np.random.RandomState(7).permutation(B)
    lZp, dlZp = dev.loglik_batch(k._kspec(), thetas[perm], grad=True)
    assert np.array_equal(lZp, lZ[perm]) and np.array_equal(dlZp, dlZ[perm]), N
    assert np.array_equal(dev.loglik_batch(k._kspec(), thetas[perm][:3], grad=False), lZv[perm][:3]), N
    out['lZ%%d' %% N], out['dlZ%%d' %% N], out['lZv%%d' %% N] = lZ, dlZ, lZv
    dev.close()
np.savez(%(path)r, **out)
print('child ok')
"""


def test_results_do_not_depend_on_group_size_or_arrangement(tmp_path):
    """The same batch under eight arrangements -- the default, groups of 5 and of 2 as one
    panel launch each, groups swept in lock-step from 2 members on, one group in flight
    instead of two, and round 5's forms of the sweep: its round-4 phases (whole-CU tile tasks,
    every trailing update a product of the tile engine), dense row panels that fold only the
    last update in while the fused task still solves its tile (full-width right-hand-side
    tiles, left-looking everywhere, R^-1 column by column), and solo launches (one
    workgroup per member) -- and, inside each, the members permuted: identical bits
    everywhere, and equal to the single evaluations. Sizes: one tile row, a 1024-block, a
    whole-matrix launch (np = 2048) and the blocked sweep above np = 4096 (np = 4224)."""
    cases = [(200, 2, 11), (1024, 8, 13), (2048, 8, 7), (4200, 4, 5)]
    envs = [{}, {'GPX_GROUP_MEMBERS': '5', 'GPX_SWEEP_MIN_MEMBERS': '99'},
            {'GPX_GROUP_MEMBERS': '2', 'GPX_SWEEP_MIN_MEMBERS': '99'},
            {'GPX_GROUP_MEMBERS': '4', 'GPX_SWEEP_MIN_MEMBERS': '2'},
            {'GPX_GROUP_INFLIGHT': '1', 'GPX_SWEEP_MIN_MEMBERS': '3'},
            {'GPX_SWEEP_MIN_MEMBERS': '2', 'GPX_SWEEP_LITE': '0'},
            {'GPX_SWEEP_MIN_MEMBERS': '2', 'GPX_SWEEP_PRE': '0', 'GPX_SWEEP_FOLD': '1',
             'GPX_SWEEP_NARROW': '0', 'GPX_SWEEP_RIGHT': '0', 'GPX_SWEEP_INVBLOCK': '1'},
            {'GPX_SOLO_MIN_MEMBERS': '2', 'GPX_SOLO_MAX_NP': '4096'}]
    res = []
    for i, e in enumerate(envs):
        path = str(tmp_path / ('g%d.npz' % i))
        code = _CHILD % dict(root=ROOT, tests=os.path.join(ROOT, 'tests'), cases=cases, path=path)
        out = run_child([sys.executable, '-c', code], env=dict(os.environ, **e), timeout=900)
        assert out.returncode == 0 and 'child ok' in out.stdout, (e, out.stderr[-3000:])
        res.append(np.load(path))
    for r in res[1:]:
        for key in res[0].files:
            assert np.array_equal(r[key], res[0][key]), key
    # ... and the single evaluations (this process)
    import pygp_amd
    from pygp_amd import _lib
    for N, D, B in cases:
        X, y, _ = recipes.synthetic(N, D)
        k = pygp_amd.kernels.SE(1.0, np.ones(D))
        thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
        dev = _lib.Handle(0)
        dev.set_data(X, y)
        for b in range(B):
            l1, d1 = _single(dev, k, thetas[b], True)
            assert l1 == res[0]['lZ%d' % N][b] and np.array_equal(d1, res[0]['dlZ%d' % N][b]), (N, b)
            assert _single(dev, k, thetas[b], False) == res[0]['lZv%d' % N][b], (N, b)
        dev.close()


def test_groups_at_the_sizes_of_the_sample_loops():
    """B = 256 thetas at N = 512, 1024 and 2048 (the regime the batched-theta consumers live
    in, and the C4s records of bench.py): EVERY member against the oracle -- lZ and the
    gradient at N = 512, lZ at N = 1024 with the gradient for every 16th member, a sample of
    members at N = 2048 (the oracle needs seconds per member there) --, members evaluated on
    their own bit for bit, and the value-only batch against the batch with gradients to
    rounding (the same bits up to N = 1024, where both take the single panel)."""
    import pygp_amd
    from pygp_amd import _lib
    D, B = 8, 256
    spec = orc.se_spec(1.0, np.ones(D))
    for N in (512, 1024, 2048):
        X, y, _ = recipes.synthetic(N, D)
        k = pygp_amd.kernels.SE(1.0, np.ones(D))
        thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
        dev = _lib.Handle(0)
        dev.set_data(X, y)
        lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
        lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
        assert np.all(np.isfinite(lZ)) and np.all(np.isfinite(dlZ))
        if N <= 1024:
            assert np.array_equal(lZv, lZ)
        else:
            nt.assert_allclose(lZv, lZ, rtol=1e-12)
        members = range(B) if N <= 1024 else (0, 77, 128, 255)
        for b in members:
            with_grad = N == 512 or b % 16 == 0 or N == 2048
            if with_grad:
                want_lZ, want_dlZ = orc.exact_eval(spec, thetas[b], X, y)
                assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * np.max(np.abs(want_dlZ)), (N, b)
            else:
                sb = orc.spec_set_hyper(orc._deepcopy_spec(spec), thetas[b][1:-1])
                R, a = orc.exact_update(sb, thetas[b][0], thetas[b][-1], X, y)
                want_lZ = orc.exact_loglik(sb, thetas[b][0], X, R, a, False)
            nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ)
        for b in (0, 77, 128, 255):
            l1, d1 = _single(dev, k, thetas[b], True)
            assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), (N, b)
            assert _single(dev, k, thetas[b], False) == lZv[b], (N, b)
        dev.close()


_CHILD_POST = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import recipes, pygp_amd
from pygp_amd import _lib
out = {}
for N, D, B, M in %(cases)r:
    X, y, Xs = recipes.synthetic(N, D, n_test=M)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    for grad in (False, True):
        parts = dev.posterior_batch(k._kspec(), thetas, Xs, grad=grad)
        for i, p in enumerate(parts):
            out['p%%d_%%d_%%d_%%d' %% (N, M, int(grad), i)] = p
    dev.close()
np.savez(%(path)r, **out)
print('child ok')
"""


def test_posterior_batch_in_groups(tmp_path):
    """gpx_posterior_batch = [m.posterior(X, grad) for m in samples] (mcmc.py:75-77) with the
    members in lock-step: cross build, solve, reductions and input gradients one launch each
    over the group. Against the oracle (1e-6, BASELINE.json north_star), and -- the cuts of
    the products depending on (np, test points) only -- bit-equal to the same call with the
    groups switched off (one context and stream per member, rounds 1-3), for few and many
    test points (block substitution / one product with the completed inverse / split-K)."""
    # (20 models at N = 1500: a lock-step sweep over a whole matrix that keeps the inverses
    # of its 1024-blocks for the block substitution)
    cases = [(333, 3, 7, 21), (1500, 2, 20, 9), (2300, 4, 5, 700), (2300, 4, 5, 130)]
    res = []
    for i, e in enumerate([{}, {'GPX_GROUP_MAX_NP': '0'}, {'GPX_GROUP_MEMBERS': '2'}]):
        path = str(tmp_path / ('p%d.npz' % i))
        code = _CHILD_POST % dict(root=ROOT, tests=os.path.join(ROOT, 'tests'), cases=cases,
                                  path=path)
        out = run_child([sys.executable, '-c', code], env=dict(os.environ, **e), timeout=900)
        assert out.returncode == 0 and 'child ok' in out.stdout, (e, out.stderr[-3000:])
        res.append(np.load(path))
    for r in res[1:]:
        for key in res[0].files:
            assert np.array_equal(r[key], res[0][key]), key
    for N, D, B, M in cases[:2]:
        X, y, Xs = recipes.synthetic(N, D, n_test=M)
        spec = orc.se_spec(1.0, np.ones(D))
        thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
        for b in range(B):
            sb = orc.spec_set_hyper(orc._deepcopy_spec(spec), thetas[b][1:-1])
            R, a = orc.exact_update(sb, thetas[b][0], thetas[b][-1], X, y)
            want = orc.exact_posterior_grad(sb, thetas[b][-1], X, R, a, Xs)
            for i, w in enumerate(want):
                nt.assert_allclose(res[0]['p%d_%d_1_%d' % (N, M, i)][b], w, rtol=1e-6, atol=1e-6)


def test_randomised_batches_against_single_evaluations():
    """tools/soak_groups.py for 12 seconds: random sizes (130 .. 4400 points), dimensions,
    eight kernel families, batch lengths 2 .. 70, value-only / with gradients on ONE handle --
    members against the same theta on its own (bit for bit), against the oracle (1e-8), and
    posteriors of a batch against the single model's. (A 4-minute run of the same script:
    3 675 batches, 66 341 members, worst lZ error 5.5e-12.)"""
    out = run_child([sys.executable, os.path.join(ROOT, 'tools', 'soak_groups.py'), '12', '7'],
                    timeout=600)
    assert out.returncode == 0 and 'soak ok' in out.stdout, out.stderr[-3000:]


def test_randomised_batches_from_one_point_up():
    """The same script over every kernel family of tests/recipes.py and sizes from ONE point
    (the reference's own demos and tests: N = 5 ... 200) for 8 seconds. (150 s: 14 824
    batches, 295 151 members, worst lZ error 6.1e-12.)"""
    out = run_child([sys.executable, os.path.join(ROOT, 'tools', 'soak_groups.py'), '8', '5',
                     'small'], timeout=600)
    assert out.returncode == 0 and 'soak ok' in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_randomised_walks_over_the_model_state_machine():
    """tools/soak_model.py for 10 seconds: add_data in chunks of random length (first data,
    in-place appends, new 128-blocks, past the capacity), set_hyper after appends, copies
    that keep appending, resets -- every state against the oracle from scratch on the data
    the model holds, log-likelihood, gradient, posterior and its input gradients. (120 s:
    543 walks, 1 904 states, worst lZ 2.2e-13 relative, posterior 3.0e-13.)"""
    out = run_child([sys.executable, os.path.join(ROOT, 'tools', 'soak_model.py'), '10', '3'],
                    timeout=600)
    assert out.returncode == 0 and 'soak ok' in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_long_batches_bad_thetas_and_recovery():
    """A batch much longer than the groups in flight (3 000 thetas at N = 200: a dozen groups
    per slot) equals the same thetas in short batches; a non-finite theta in the middle of a
    batch is an error (RuntimeError from the C ABI, like non-finite hypers of a single
    model) that leaves nothing running, and the handle evaluates the next batch as if
    nothing had happened."""
    import pygp_amd
    from pygp_amd import _lib
    N, D, B = 200, 2, 3000
    X, y, _ = recipes.synthetic(N, D)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    rng = np.random.RandomState(5)
    thetas = recipes.theta0(D) + 0.2 * rng.randn(B, D + 3)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    assert np.all(np.isfinite(lZ)) and np.all(np.isfinite(dlZ))
    for lo in (0, 1234, 2990):
        l2, d2 = dev.loglik_batch(k._kspec(), thetas[lo:lo + 10], grad=True)
        assert np.array_equal(l2, lZ[lo:lo + 10]) and np.array_equal(d2, dlZ[lo:lo + 10])
    bad = thetas[:700].copy()
    bad[611, 0] = np.nan
    with pytest.raises(_lib.GpxError):
        dev.loglik_batch(k._kspec(), bad, grad=False)
    again = dev.loglik_batch(k._kspec(), thetas[:700], grad=True)
    assert np.array_equal(again[0], lZ[:700]) and np.array_equal(again[1], dlZ[:700])
    dev.close()


@pytest.mark.parametrize('N,D,B', [(1, 1, 3), (2, 2, 5), (5, 1, 40), (20, 1, 7), (100, 3, 33),
                                   (128, 2, 4), (129, 2, 4)])
def test_tiny_datasets_in_groups(N, D, B):
    """The sizes of the reference's own demos and tests (N = 5 ... 200, one tile): a group of
    members whose matrix is a single 128-tile (the batched leaf, no panel launch) and the
    first size with two tiles; against the oracle, against single evaluations bit for bit,
    and the posteriors of a batch against the single model's."""
    from pygp_amd import _lib
    X, y, Xs = recipes.synthetic(N, D, n_test=6)
    desc = ('se', (1.0, list(np.linspace(0.5, 1.5, D))), {})
    k, thetas = _thetas(desc, D, B)
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    spec = oracle_spec(desc)
    for b in range(B):
        want_lZ, want_dlZ = orc.exact_eval(spec, thetas[b], X, y)
        nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ, atol=1e-12)
        nt.assert_allclose(lZv[b], want_lZ, rtol=RTOL_LZ, atol=1e-12)
        assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * max(1.0, np.max(np.abs(want_dlZ)))
        l1, d1 = _single(dev, k, thetas[b], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
        assert _single(dev, k, thetas[b], False) == lZv[b], b
    mu, s2 = dev.posterior_batch(k._kspec(), thetas, Xs)
    for b in (0, B - 1):
        kb = k.copy(thetas[b][1:-1])
        dev.exact_update(kb._kspec(), thetas[b][0], thetas[b][-1])
        m1, v1 = dev.exact_posterior(Xs)
        nt.assert_allclose(mu[b], m1, rtol=1e-12, atol=1e-13)
        nt.assert_allclose(s2[b], v1, rtol=1e-10, atol=1e-13)
        sb = orc.spec_set_hyper(orc._deepcopy_spec(spec), thetas[b][1:-1])
        R, a = orc.exact_update(sb, thetas[b][0], thetas[b][-1], X, y)
        wm, wv = orc.exact_posterior(sb, thetas[b][-1], X, R, a, Xs)[:2]
        nt.assert_allclose(mu[b], wm, rtol=1e-6, atol=1e-9)
        nt.assert_allclose(s2[b], wv, rtol=1e-6, atol=1e-9)
    dev.close()


def test_one_handle_through_changing_datasets_and_batch_sizes():
    """One handle, datasets that grow and shrink across the one-tile / two-tile / panel /
    multi-block boundaries, batches of 0, 1, 2 and a few hundred thetas, one and a few hundred
    test points, with input gradients: the groups' workspaces are re-reserved for every shape
    and nothing of the previous one may leak into the next. Against the oracle."""
    from pygp_amd import _lib
    dev = _lib.Handle(0)
    D = 2
    desc = ('se', (1.0, [0.7, 1.3]), {})
    spec = oracle_spec(desc)
    rng = np.random.RandomState(11)
    for step, (N, B, M) in enumerate([(20, 300, 1), (700, 2, 300), (5, 1, 3), (1500, 9, 40),
                                      (128, 0, 2), (128, 257, 130), (2100, 3, 1), (1, 130, 5)]):
        X, y, Xs = recipes.synthetic(N, D, n_test=M, seed=step)
        k, _ = _thetas(desc, D, 1)
        base = np.r_[np.log(0.1), k.get_hyper(), 0.05]
        thetas = base + 0.1 * rng.randn(B, base.size)
        dev.set_data(X, y)
        lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
        lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
        assert lZ.shape == (B,) and dlZ.shape == (B, base.size) and lZv.shape == (B,)
        out = dev.posterior_batch(k._kspec(), thetas, Xs, grad=True)
        assert out[0].shape == (B, M) and out[2].shape == (B, M, D)
        for b in sorted(set([0, B // 2, B - 1]) & set(range(B))):
            want_lZ, want_dlZ = orc.exact_eval(spec, thetas[b], X, y)
            nt.assert_allclose(lZ[b], want_lZ, rtol=RTOL_LZ, atol=1e-12, err_msg=str((step, b)))
            assert lZv[b] == lZ[b] or abs(lZv[b] - want_lZ) <= RTOL_LZ * abs(want_lZ) + 1e-12
            assert np.max(np.abs(dlZ[b] - want_dlZ)) <= 1e-7 * max(1.0, np.max(np.abs(want_dlZ)))
            sb = orc.spec_set_hyper(orc._deepcopy_spec(spec), thetas[b][1:-1])
            R, a = orc.exact_update(sb, thetas[b][0], thetas[b][-1], X, y)
            want = orc.exact_posterior_grad(sb, thetas[b][-1], X, R, a, Xs)
            for got, w in zip(out, want):
                nt.assert_allclose(got[b], w, rtol=1e-6, atol=1e-7, err_msg=str((step, b)))
    dev.close()


def test_groups_above_np_16384():
    """Two thetas at N = 17 000 (np = 17 024: above the sizes of BASELINE.json; groups run up to
    np = 32 768 since the end of round 4): a group of two in lock-step, bit-equal to the single
    evaluations, whose path is oracle-checked at N = 16 384 (tests/test_gpu_gp.py)."""
    import pygp_amd
    from pygp_amd import _lib
    N, D, B = 17000, 4, 2
    X, y, _ = recipes.synthetic(N, D)
    k = pygp_amd.kernels.SE(1.0, np.linspace(0.6, 1.4, D))
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(B)])
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    assert dev.batch_plan(B, True)['arrangement'].startswith('groups')
    lZ, dlZ = dev.loglik_batch(k._kspec(), thetas, grad=True)
    lZv = dev.loglik_batch(k._kspec(), thetas, grad=False)
    for b in range(B):
        l1, d1 = _single(dev, k, thetas[b], True)
        assert l1 == lZ[b] and np.array_equal(d1, dlZ[b]), b
        assert _single(dev, k, thetas[b], False) == lZv[b], b
    assert np.all(np.isfinite(lZ)) and np.all(np.isfinite(dlZ))
    nt.assert_allclose(lZv, lZ, rtol=1e-12)
    dev.close()


_CHILD_FULLW = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import recipes, pygp_amd
from pygp_amd import _lib
out = {}
for N, D in %(cases)r:
    X, y, Xs = recipes.synthetic(N, D, n_test=N // 2)
    k = pygp_amd.kernels.SE(1.0, np.ones(D))
    dev = _lib.Handle(0)
    dev.set_data(X, y)
    for b in range(3):
        th = recipes.theta_sweep(D, b)
        lZ, dlZ = dev.exact_eval(k.copy(th[1:-1])._kspec(), th[0], th[-1], True)
        out['lZ%%d_%%d' %% (N, b)], out['dlZ%%d_%%d' %% (N, b)] = lZ, dlZ
    # the factorisation of the last evaluation is the handle's: a posterior at N / 2 points
    # multiplies by the R^-1 it left behind
    mu, s2 = dev.exact_posterior(Xs)
    out['mu%%d' %% N], out['s2%%d' %% N] = mu, s2
    # the same thetas as members of one panel launch (3) and of a lock-step sweep (17)
    thetas = np.array([recipes.theta_sweep(D, b) for b in range(17)])
    for B in (3, 17):
        lZ, dlZ = dev.loglik_batch(k._kspec(), thetas[:B], grad=True)
        for b in range(3):
            assert lZ[b] == out['lZ%%d_%%d' %% (N, b)] and np.array_equal(dlZ[b], out['dlZ%%d_%%d' %% (N, b)]), (N, B, b)
    dev.close()
np.savez(%(path)r, **out)
print('child ok')
"""


def test_inverse_assembled_inside_the_launch_against_trtri_behind_it(tmp_path):
    """Evaluations with gradients up to np = 4096 assemble ALL of R^-1 inside the whole-matrix
    launch (chunked sums as worker tasks, gpx_grad_full_w); GPX_GRAD_FULL_W=0 leaves the
    completion to gpx_trtri behind the launch. Both against the oracle
    (/root/reference/pygp/inference/exact.py:118-141), lZ with identical bits (R and a do not
    depend on the inverse), gradients and the posterior that reuses the inverse to rounding of
    each other; in each, members of one panel launch and of a lock-step sweep (the same sums
    as ONE product on the tile engine) bit-equal to the single evaluations. N = 1300 pads to
    11 tiles (bulk chunks of 4 + 4 + 1 tile rows and the last row), N = 2048 is 16, N = 2500
    is 20 tiles: from 17 on the early bulk chunks are 128x128 tasks."""
    cases = [(1300, 3), (2048, 8), (2500, 4)]
    res = []
    for e in ({}, {'GPX_GRAD_FULL_W': '0'}):
        path = str(tmp_path / ('w%d.npz' % len(res)))
        code = _CHILD_FULLW % dict(root=ROOT, tests=os.path.join(ROOT, 'tests'), cases=cases, path=path)
        out = run_child([sys.executable, '-c', code], env=dict(os.environ, **e), timeout=600)
        assert out.returncode == 0 and 'child ok' in out.stdout, (e, out.stderr[-3000:])
        res.append(np.load(path))
    for N, D in cases:
        X, y, Xs = recipes.synthetic(N, D, n_test=N // 2)
        spec = orc.se_spec(1.0, np.ones(D))
        for b in range(3):
            th = recipes.theta_sweep(D, b)
            want_lZ, want_dlZ = orc.exact_eval(spec, th, X, y)
            for r in res:
                nt.assert_allclose(r['lZ%d_%d' % (N, b)], want_lZ, rtol=RTOL_LZ)
                assert np.max(np.abs(r['dlZ%d_%d' % (N, b)] - want_dlZ)) <= 1e-8 * np.max(np.abs(want_dlZ))
            assert res[0]['lZ%d_%d' % (N, b)] == res[1]['lZ%d_%d' % (N, b)], (N, b)
            nt.assert_allclose(res[0]['dlZ%d_%d' % (N, b)], res[1]['dlZ%d_%d' % (N, b)],
                               rtol=0, atol=1e-11 * np.max(np.abs(want_dlZ)))
        th = recipes.theta_sweep(D, 2)
        sb = orc.spec_set_hyper(orc._deepcopy_spec(spec), th[1:-1])
        R, a = orc.exact_update(sb, th[0], th[-1], X, y)
        wm, ws = orc.exact_posterior(sb, th[-1], X, R, a, Xs)
        for r in res:
            nt.assert_allclose(r['mu%d' % N], wm, rtol=0, atol=1e-6)
            nt.assert_allclose(r['s2%d' % N], ws, rtol=0, atol=1e-6)
        nt.assert_allclose(res[0]['s2%d' % N], res[1]['s2%d' % N], rtol=0, atol=1e-11)
