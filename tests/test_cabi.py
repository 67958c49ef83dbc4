"""CPU-only checks of the C-ABI boundary: libgpx.so builds/loads, exports every
symbol include/gpx.h declares, and the Python binding table matches the
header. No compute is called (there is no GPU here)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'gpx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gpx_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def libpath():
    from pygp_amd import build
    return build.build(verbose=False)


def _gpu_present():
    """Device count through the library itself (no torch in this file)."""
    from pygp_amd import _lib
    try:
        return _lib.device_count() > 0
    except _lib.GpxError:
        return False


def test_header_declares_functions():
    names = header_functions()
    assert 'gpx_exact_eval' in names and 'gpx_kernel_get' in names
    assert len(names) >= 20


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    missing = [n for n in header_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header(libpath):
    from pygp_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    assert _lib.lib().gpx_version() == 100


def test_fails_loudly_without_gpu():
    """The product path has no CPU fallback: without a device it must raise."""
    import numpy as np
    import pygp_amd
    from pygp_amd import _lib
    if _gpu_present():
        pytest.skip('a GPU is present')
    k = pygp_amd.kernels.SE(1.0, [1.0, 1.0])
    with pytest.raises(_lib.GpxError):
        k.get(np.zeros((3, 2)))
    gp = pygp_amd.BasicGP(.1, 1, .1)
    with pytest.raises(_lib.GpxError):
        gp.add_data(np.zeros((3, 1)), np.zeros(3))


def test_product_never_imports_oracle():
    """pygp_amd must not reach into oracle/ (the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, 'pygp_amd')
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r'^\s*(from|import)\s+\.*oracle', src, re.M), f
                assert 'gp_oracle' not in src, f


def test_partition_rule_is_the_same_in_c_and_python(libpath):
    """gpx_batch_partition (the in-library multi-device entry) and
    pygp_amd.batch.partition (torchrun launches) deal members identically."""
    from pygp_amd import _lib
    from pygp_amd.batch import partition
    for B in (0, 1, 5, 8, 63, 64, 65, 1000):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for rank in range(world):
                lo, hi = _lib.batch_partition(B, world, rank)
                assert (lo, hi) == partition(B, world, rank)
                cover += list(range(lo, hi))
            assert cover == list(range(B))


def test_multi_device_entry_refuses_missing_devices(libpath):
    """ndev beyond the devices present (here: none) is a clean error, not a crash."""
    import numpy as np
    import pygp_amd
    from pygp_amd import _lib
    if _gpu_present():
        pytest.skip('a GPU is present')
    k = pygp_amd.kernels.SE(1.0, [1.0, 1.0])
    with pytest.raises(_lib.GpxError):
        _lib.loglik_batch_multi(k._kspec(), np.zeros((2, 5)), np.zeros((4, 2)), np.zeros(4),
                                ndev=2)
    with pytest.raises(_lib.GpxError):
        _lib.posterior_batch_multi(k._kspec(), np.zeros((2, 5)), np.zeros((3, 2)),
                                   np.zeros((4, 2)), np.zeros(4), ndev=2)


@pytest.mark.parametrize('ndev', [1, 2, 3, 8])
@pytest.mark.parametrize('B', [0, 1, 5, 64, 65])
@pytest.mark.parametrize('width', [2, 13])
def test_multi_device_pack_and_scatter(libpath, ndev, B, width):
    """The host halves of the in-library gather (multi.hip): every device packs its
    block of member rows into a NaN-padded slot, the slots are concatenated in device
    order (what ncclAllGather delivers) and scattered back by the partition rule.
    Covers devices without members (B < ndev), ragged blocks, and both row widths of
    gpx_loglik_batch_multi (value only: [lZ | info]; with gradients: [lZ | dlZ | info])."""
    import numpy as np
    from pygp_amd import _lib
    slot = _lib.lib().gpx_multi_slot(B, ndev)
    assert slot == -(-B // ndev)
    rows = np.arange(B * width, dtype=float).reshape(B, width) + 0.5
    image = []
    for dev in range(ndev):
        lo, hi = _lib.batch_partition(B, ndev, dev)
        assert hi - lo <= slot
        pack = _lib.multi_pack(rows[lo:hi], slot)
        assert pack.shape == (slot, width)
        assert np.array_equal(pack[:hi - lo], rows[lo:hi])
        assert np.all(np.isnan(pack[hi - lo:]))            # padding is never a number
        image.append(pack)
    gathered = np.concatenate(image) if slot else np.zeros((0, width))
    out = _lib.multi_scatter(gathered, B, ndev, width)
    assert out.shape == (B, width)
    assert np.array_equal(out, rows)                        # no NaN leaks, order kept
    if B:
        with pytest.raises(_lib.GpxError):
            _lib.multi_pack(rows, max(0, B - 1))            # more rows than the slot holds


@pytest.mark.parametrize('stream', [True, False])
def test_panel_task_graph_is_a_valid_schedule(stream):
    """The diagonal-panel kernel (pygp_amd/csrc/panel.hip) runs a host-built task list:
    workgroups claim tasks in list order and wait on device counters, so the list must
    be a topological order of those counter dependencies or the launch would stall
    until its timeout. gpx_panel_graph_check replays the list on the host for every
    block size and worker count the driver uses, for the round-2 graph (row panels
    solved beside the streamed leaf) and the round-1 graph."""
    from pygp_amd import _lib
    sizes = {}
    for T in range(2, 9):
        for workers in (32, 64, 128):
            n = _lib.panel_graph_check(T, workers, stream)
            assert sizes.setdefault(T, n) == n          # the graph does not depend on workers
    assert sizes[2] < sizes[4] < sizes[8] <= 1024       # the trace buffer holds 1024 tasks
    with pytest.raises(RuntimeError):
        _lib.panel_graph_check(1, 64, stream)
    with pytest.raises(RuntimeError):
        _lib.panel_graph_check(9 if not stream else 33, 64, stream)


def test_whole_matrix_task_graph_is_a_valid_schedule():
    """Round 3: a value-only factorisation of a small matrix is ONE panel launch over all
    its tiles (up to 32 of them: the chain of diagonal tiles, every trailing update of
    every step, and the inverses of the 1024-blocks the blocked driver would have left
    behind). Same replay: the list is a permutation and a topological order, every tile
    ends at the counter value a finished tile stands at, for the worker counts in use."""
    from pygp_amd import _lib
    last = 0
    for T in (9, 12, 16, 20, 25, 32):
        for workers in (128, 160, 250):
            n = _lib.panel_graph_check(T, workers, True)
            # with one more tile column right of the matrix: the right-hand side of the
            # forward substitution, solved by the same launch
            m = _lib.panel_graph_check(T, workers, True, extra=1)
            assert m > n
        assert n > last
        last = n
    assert last < 32000                                  # task ids and counters are shorts
    with pytest.raises(RuntimeError):
        _lib.panel_graph_check(16, 160, True, extra=2)   # one right-hand-side column only
    # round 4: matrices of at most 8 tiles (one panel) take their right-hand side along too
    for T in range(2, 9):
        for workers in (32, 96, 128):
            assert _lib.panel_graph_check_rhs(T, workers) > _lib.panel_graph_check(T, workers)
    assert _lib.panel_graph_check_rhs(12, 160) == _lib.panel_graph_check(12, 160, True, extra=1)
    # round 5: the launch of an evaluation with gradients assembles ALL of R^-1 (chunked sums)
    for T in (12, 16, 17, 24, 31, 32):
        for workers in (41, 125, 241):
            assert _lib.panel_graph_check_full(T, workers) != _lib.panel_graph_check_rhs(T, workers)
    assert _lib.panel_graph_check_full(8, 96) == _lib.panel_graph_check_rhs(8, 96)


def test_wide_panel_task_graph_is_a_valid_schedule():
    """Wide panel launches (round 3): the block's tiles plus E tile columns to the right,
    whose row-panel tiles are solved beside the leaves and whose diagonal block takes the
    block's update inside the launch. Same replay for every shape the driver uses (the
    top-level chain at N <= 8192: T = E = 8 and ragged last blocks; inside 2048-blocks)."""
    from pygp_amd import _lib
    plain = _lib.panel_graph_check(8, 64)
    for T, E in ((8, 8), (8, 1), (8, 5), (2, 8), (4, 4), (3, 8)):
        for workers in (64, 96):
            n = _lib.panel_graph_check(T, workers, extra=E)
            assert n > _lib.panel_graph_check(T, workers)
    assert _lib.panel_graph_check(8, 96, extra=8) < 4096 and plain < 1024   # trace buffers
    with pytest.raises(RuntimeError):
        _lib.panel_graph_check(8, 64, extra=9)


def test_lockstep_sweep_is_consistent_with_the_task_graph():
    """Groups of 16 members or more factor their diagonal blocks phase by phase over all
    members (sweep_block in chol.hip, round 4): F(0), then per tile row the left-looking
    updates on the tile engine and the row-panel phase. gpx_sweep_check replays that order
    against the counter thresholds of the panel launch's own task graph -- what a task would
    have waited for is there when its phase runs, every product multiplies final row panels,
    every tile ends where a finished tile stands -- for 1024-blocks, their ragged cousins and
    whole matrices with the right-hand side riding along."""
    from pygp_amd import _lib
    for T in range(1, 9):
        _lib.sweep_check(T, False)
    for T in (2, 3, 8, 9, 12, 16, 25, 32):
        _lib.sweep_check(T, True)
    with pytest.raises(RuntimeError):
        _lib.sweep_check(33, False)


def test_dense_row_panels_of_the_sweep_are_consistent_with_the_task_graph():
    """Round 5: the tiles right of (s, s+1) are solved by a dense launch (sweep_xs_kernel, two
    workgroups a CU, tile in registers) whose tasks apply the last `depth` trailing updates of
    their tile themselves; earlier steps stay one product of the tile engine, tile (s, s+1)
    keeps the fused task. gpx_sweep_check_lite replays that order against the task graph's
    counters for every depth: each folded step multiplies final row panels, each dense task
    finds its tile exactly at the step where its own updates start and the leaf of its row
    done, every tile ends where a finished tile stands."""
    from pygp_amd import _lib
    for T in range(1, 9):
        for depth in (-1, 0, 1, 3, T):
            _lib.sweep_check_lite(T, False, depth)
    for T in (2, 3, 4, 8, 9, 12, 16, 25, 32):
        for depth in (-1, 0, 2, 4, T):
            _lib.sweep_check_lite(T, True, depth)
    # the right-looking form of small matrices (depth -2): every dense launch applies ONE step
    # to all tiles from its row down
    for T in range(1, 9):
        _lib.sweep_check_lite(T, True, -2)
        _lib.sweep_check_lite(T, False, -2)
    with pytest.raises(RuntimeError):
        _lib.sweep_check_lite(33, False, -1)


def test_solo_lists_are_sequential_orders():
    """Round 5, an opt-in arrangement of groups: one workgroup per member runs the member's whole
    task graph in GENERATION order (gpx_panel_solo). That order must be sequentially valid --
    counters and polled data alike come from earlier tasks -- for single panels with a
    right-hand side, whole matrices, the whole inverse inside, and value-only lists (the tasks of
    the inverse left out, which nothing else waits for)."""
    from pygp_amd import _lib
    for T in range(2, 9):
        n_all = _lib.panel_solo_check(T, aug=True)
        n_val = _lib.panel_solo_check(T, aug=True, value_only=True)
        assert 0 < n_val < n_all
        _lib.panel_solo_check(T, aug=False)
    for T in (9, 12, 16, 24, 32):
        assert _lib.panel_solo_check(T, aug=True, full=True) > _lib.panel_solo_check(T, aug=True)
        _lib.panel_solo_check(T, aug=True, value_only=True)
    with pytest.raises(RuntimeError):
        _lib.panel_solo_check(33)


def test_panel_launch_co_residency_rule(libpath):
    """Every workgroup of a panel launch holds a whole CU and the progress argument needs all
    spine workgroups plus one worker resident together: the rule the launch applies
    -- 3 / 2 / 1 spine workgroups per member by members, fewer when they would not fit -- as a
    host function; what cannot fit at all is an error, not a launch that waits out its bound."""
    from pygp_amd import _lib
    assert _lib.panel_grid_check(1, 256) == (3, 247)
    assert _lib.panel_grid_check(6, 256) == (3, 232)         # the headline's group of six
    assert _lib.panel_grid_check(16, 256) == (3, 202)
    assert _lib.panel_grid_check(40, 256) == (2, 170)
    assert _lib.panel_grid_check(100, 256) == (1, 150)
    assert _lib.panel_grid_check(255, 256) == (1, 8)          # 255 spines + a worker's CU
    with pytest.raises(_lib.GpxError):
        _lib.panel_grid_check(256, 256)                       # GPX_GROUP_MEMBERS=256, sweep off
    for nmem in range(1, 256):
        sp, wk = _lib.panel_grid_check(nmem, 256)
        assert 1 <= sp <= 3 and nmem * sp + 1 <= 256 and wk >= 8
    # a smaller part: the spines give way first
    assert _lib.panel_grid_check(16, 40) == (2, 218)
    assert _lib.panel_grid_check(16, 20)[0] == 1
