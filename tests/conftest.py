import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


# A handle that switches to safe mode (another order of arithmetic) announces it with a
# RuntimeWarning whose text starts with "pygp_amd:". Nothing in the suite may meet one
# silently: it is an error in every test, and in every child process a test starts; the one
# test of the switch itself records the warning inside its child
# (test_safe_mode_against_the_oracle_and_the_automatic_switch, tools/check_safe_mode.py auto).
PYGP_WARNING_FILTER = 'error:pygp_amd:RuntimeWarning'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('filterwarnings', PYGP_WARNING_FILTER)


def pytest_collection_modifyitems(config, items):
    """A GPU test that is still running after 7 minutes (the longest takes one) is stuck in a
    device call no Python-level signal reaches: pytest-timeout's thread method (where the
    plugin is installed) dumps every thread's stack and ends the run instead of leaving it to
    the caller's own limit. (Twice in round 4 a run stopped inside gpx_destroy, in
    hipStreamDestroy of a CU-masked stream -- DESIGN.md section 4; the library creates none by
    default any more.)"""
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for item in items:
        if item.get_closest_marker('gpu') and not item.get_closest_marker('timeout'):
            item.add_marker(pytest.mark.timeout(420, method='thread'))


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip('fixture %s not generated' % name)
    return np.load(path)


@pytest.fixture(scope='session')
def g_small():
    return load_golden('g_small.npz')


@pytest.fixture(scope='session')
def g_mid():
    return load_golden('g_mid.npz')


def run_child(argv, env=None, timeout=300, **kw):
    """subprocess.run(capture_output=True, text=True) for the GPU tests that need a fresh
    process (switches read once per process), with an end the suite always reaches: a child
    still running at `timeout` gets SIGABRT (PYTHONFAULTHANDLER=1: it dumps the Python stack
    of every thread first), then SIGKILL, and the test FAILS with what the child wrote -- a
    child the kernel cannot reap is left behind rather than waited for."""
    import signal
    import subprocess
    import tempfile
    env = dict(os.environ if env is None else env, PYTHONFAULTHANDLER='1')
    env.setdefault('PYTHONWARNINGS', PYGP_WARNING_FILTER)
    with tempfile.TemporaryFile('w+') as fo, tempfile.TemporaryFile('w+') as fe:
        p = subprocess.Popen(argv, env=env, stdout=fo, stderr=fe, text=True, **kw)
        rc = None
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            for sig in (signal.SIGABRT, signal.SIGKILL):
                p.send_signal(sig)
                try:
                    p.wait(timeout=15)
                    break
                except subprocess.TimeoutExpired:
                    pass
        fo.seek(0)
        fe.seek(0)
        out, err = fo.read(), fe.read()
    if rc is None:
        pytest.fail('child still running after %d s (state now: %s)\n--- stdout\n%s\n--- stderr\n%s'
                    % (timeout, p.poll(), out[-3000:], err[-6000:]), pytrace=False)
    return subprocess.CompletedProcess(argv, rc, out, err)
