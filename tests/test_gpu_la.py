"""GPU tests of the dense fp64 engine (gemm_f64.hip, chol.hip) through the
C-ABI, against NumPy/SciPy on the same inputs."""

import numpy as np
import numpy.testing as nt
import scipy.linalg as sla
import pytest

from conftest import run_child

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from pygp_amd import _lib
    return _lib.Handle(0)


@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('shape', [(16, 16, 4), (200, 150, 70), (256, 384, 128),
                                   (129, 257, 513)])
def test_gemm(dev, ta, tb, shape):
    M, N, K = shape
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    # asymmetric operands: a transposed or row/col-swapped MFMA map cannot pass
    A = rng.randn(K, M) if ta else rng.randn(M, K)
    B = rng.randn(N, K) if tb else rng.randn(K, N)
    C0 = rng.randn(M, N)
    ref = 0.7 * (A.T if ta else A) @ (B.T if tb else B) - 1.3 * C0
    out = dev.la_gemm(A, B, ta=ta, tb=tb, alpha=0.7, beta=-1.3, Cin=C0)
    scale = np.abs(A).sum(0 if ta else 1).max() * np.abs(B).max()
    nt.assert_allclose(out, ref, rtol=0, atol=1e-13 * scale)
    out = dev.la_gemm(A, B, ta=ta, tb=tb)
    nt.assert_allclose(out, (A.T if ta else A) @ (B.T if tb else B), rtol=0,
                       atol=1e-13 * scale)


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric B (guide: catches a swapped C/D map)."""
    n = 128
    B = np.arange(n * n, dtype=float).reshape(n, n)
    nt.assert_array_equal(dev.la_gemm(np.eye(n), B), B)
    nt.assert_array_equal(dev.la_gemm(B, np.eye(n)), B)
    nt.assert_array_equal(dev.la_gemm(B, np.eye(n), ta=True), B.T)
    nt.assert_array_equal(dev.la_gemm(np.eye(n), B, tb=True), B.T)


def spd(n, seed, cond=1e4):
    rng = np.random.RandomState(seed)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    ev = np.logspace(0, np.log10(cond), n)
    return (Q * ev) @ Q.T


@pytest.mark.parametrize('n', [1, 5, 128, 200, 384, 640, 896, 1024, 1100, 2100])
def test_potrf_inverse(dev, n):
    A = spd(n, n)
    R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
    Rref = sla.cholesky(A)
    nt.assert_allclose(R, Rref, rtol=1e-10, atol=1e-12)
    assert np.all(np.tril(R, -1) == 0)
    # backward errors at fp64 level
    nA = np.linalg.norm(A)
    assert np.linalg.norm(R.T @ R - A) / nA < 1e-14 * n
    assert np.linalg.norm(Rinv @ Rref - np.eye(n)) < 1e-10 * n
    assert np.all(np.tril(Rinv, -1) == 0)
    nt.assert_allclose(Ainv, np.linalg.inv(A), rtol=1e-8, atol=1e-10)
    nt.assert_array_equal(Ainv, Ainv.T)


@pytest.mark.parametrize('cond', [1e6, 1e10])
def test_potrf_backward_error_ill_conditioned(dev, cond):
    """Backward stability of the panel path (row panels by blocked forward substitution
    over the leaf's 16x16 inverses, diagonal update from LDS) does not depend on the
    condition number: ||R^T R - A|| / ||A|| stays at the level LAPACK reaches on the same
    matrix, and A A^-1 - I within a small multiple of LAPACK's."""
    n = 1024
    A = spd(n, 7, cond)
    A = (A + A.T) / 2
    R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
    Rl = sla.cholesky(A)
    nA = np.linalg.norm(A)
    be, bel = np.linalg.norm(R.T @ R - A) / nA, np.linalg.norm(Rl.T @ Rl - A) / nA
    assert be < 8 * bel and be < 2e-15
    res = np.linalg.norm(A @ Ainv - np.eye(n))
    resl = np.linalg.norm(A @ sla.cho_solve((Rl, False), np.eye(n)) - np.eye(n))
    assert res < 8 * resl


def test_potrf_multi_block_driver_ill_conditioned(dev):
    """The same through the multi-block driver (n = 4096: four 1024-blocks, chol.hip):
    above the first block every row panel is an explicit-inverse product
    R[k, k+1:] = W_kk^T A[k, k+1:] where the reference substitutes (dpotrf + dtrtrs,
    /root/reference/pygp/inference/exact.py:54-55,88). The backward error does not
    depend on the condition number (an inverse that lost accuracy with cond(R_kk) would
    show here as 1e-11 at cond 1e10) and stays at a few eps; its constant is that of
    any right-looking rank-nb sweep, which rounds the trailing matrix once per block
    step -- measured 10-14x OpenBLAS dpotrf's 1e-16 at this size, where a NumPy
    emulation of the same sweep WITH substitution gives 5x and the explicit inverse
    adds at most 1.4x (profiles/r03_cond_backward_err.txt). The
    residual of the symmetric inverse stays within 8x LAPACK's."""
    n = 4096
    eye = np.eye(n)
    bes = {}
    for cond in (1e6, 1e10):
        A = spd(n, 11, cond)
        A = (A + A.T) / 2
        R, Rinv, Ainv = dev.la_potrf(A, inverse=True)
        Rl = sla.cholesky(A)
        nA = np.linalg.norm(A)
        be, bel = np.linalg.norm(R.T @ R - A) / nA, np.linalg.norm(Rl.T @ Rl - A) / nA
        bes[cond] = be
        assert be < 4e-15 and be < 20 * bel, (cond, be, bel)
        res = np.linalg.norm(A @ Ainv - eye)
        resl = np.linalg.norm(A @ sla.cho_solve((Rl, False), eye) - eye)
        assert res < 8 * resl, (cond, res, resl)
        ri = np.linalg.norm(Rinv @ R - eye)              # R^-1 against its own factor
        ril = np.linalg.norm(sla.solve_triangular(Rl, eye) @ Rl - eye)
        assert ri < 8 * ril + 1e-12 * n, (cond, ri, ril)
    assert bes[1e10] < 2 * bes[1e6]          # no dependence on the condition number


def test_potrf_not_positive_definite(dev):
    A = spd(300, 1)
    A[200, 200] = -1.0
    with pytest.raises(np.linalg.LinAlgError):
        dev.la_potrf(A)
    # the handle stays usable
    R = dev.la_potrf(spd(64, 2))
    nt.assert_allclose(R, sla.cholesky(spd(64, 2)), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('panel,stream', [('1024', '1'), ('1024', '0'), ('0', '1')])
def test_panel_kernel_switch(panel, stream):
    """The bottom-level paths -- the panel launch (pygp_amd/csrc/panel.hip) with its
    round-2 task graph (row panels solved beside the streamed leaf, default) and with
    the round-1 graph (GPX_PANEL_STREAM=0), and the recursion down to the leaves
    (GPX_PANEL=0) -- factor and invert diagonal blocks of 2..11 tiles; the switches are
    read once per process, so the probe runs in a child."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPX_PANEL=panel, GPX_PANEL_STREAM=stream,
               GPX_PANEL_TIMEOUT_MS='500')
    out = run_child([sys.executable, os.path.join(root, 'tools', 'panel_dbg.py'),
                          '256', '640', '1024', '1300'], env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split() for l in out.stdout.splitlines() if 'R err' in l]
    assert len(rows) == 4
    for r in rows:
        assert float(r[4]) < 1e-12 and float(r[7]) < 1e-11 and float(r[10]) < 1e-11


def test_wide_panel_switch():
    """GPX_PANEL_WIDE=1 / 2 (off by default: measured slower): the panel launch of a block
    also solves the row-panel block to its right and updates the next diagonal block, its
    tasks on those tiles waiting for gate counters that the trailing launches of the step
    before move from another stream. Factor, inverse and symmetric inverse stay within
    the tolerances of the default path for 2 .. 5 blocks (a child per setting: the switch
    is read once per process)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for level in ('1', '2'):
        env = dict(os.environ, GPX_PANEL_WIDE=level, GPX_PANEL_TIMEOUT_MS='1000')
        out = run_child([sys.executable, os.path.join(root, 'tools', 'panel_dbg.py'),
                              '2048', '2304', '4096', '5000'], env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        rows = [l.split() for l in out.stdout.splitlines() if 'R err' in l]
        assert len(rows) == 4
        for r in rows:
            assert float(r[4]) < 1e-12 and float(r[7]) < 1e-11 and float(r[10]) < 1e-11


def test_panel_kernel_strict_handoffs():
    """The panel kernel hands tiles between workgroups with agent-scope (sc1) accesses
    instead of release / acquire fences (panel.hip: the invariant is written next to
    agent_load2). GPX_PANEL_STRICT=1 adds the fences the memory model asks for; if the
    invariant holds the factor, its inverse and the symmetric inverse come out bit for
    bit the same. Each mode runs in a child (the switch is read once per process)."""
    import os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from pygp_amd import _lib\n"
        "dev = _lib.Handle(0)\n"
        "for n in (640, 1024, 2304):\n"
        "    rng = np.random.RandomState(n)\n"
        "    Q, _ = np.linalg.qr(rng.randn(n, n))\n"
        "    A = (Q * np.logspace(0, 2, n)) @ Q.T\n"
        "    A = (A + A.T) / 2\n"
        "    for rep in range(6):\n"
        "        print('start', n, rep, file=sys.stderr, flush=True)\n"
        "        R, Rinv, Ainv = dev.la_potrf(A, inverse=True)\n"
        "        print(n, hashlib.sha256(R.tobytes() + Rinv.tobytes() + Ainv.tobytes()).hexdigest())\n"
        "print('done', file=sys.stderr, flush=True)\n"
    ) % root
    outs = []
    for strict in ('0', '1'):
        env = dict(os.environ, GPX_PANEL='1024', GPX_PANEL_STRICT=strict,
                   GPX_PANEL_TIMEOUT_MS='1000')
        out = run_child([sys.executable, '-c', code], env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out.stdout.strip().splitlines())
    assert len(outs[0]) == 18 and outs[0] == outs[1]
    for n in range(3):                       # and run to run
        assert len(set(outs[0][6 * n:6 * n + 6])) == 1
