"""
Build libgpx.so (the HIP/C-ABI library) in-tree for gfx950.

    python -m pygp_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting pygp_amd/libgpx.so travels with the repository snapshot.
"""

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libgpx.so')
SOURCES = ['gpx_api.hip', 'kmat.hip', 'gemm_f64.hip', 'chol.hip', 'leaf.hip', 'panel.hip', 'vec.hip',
           'multi.hip', 'group.hip']
HEADERS = [os.path.join(CSRC, 'gpx_internal.h'), os.path.join(CSRC, 'gemm_tile.h'),
           os.path.join(CSRC, 'leaf_dev.h'),
           os.path.join(HERE, '..', 'include', 'gpx.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall',
         '-Wno-unused-function']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; libgpx.so cannot be built')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (s, r.stderr))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return o

    if jobs:
        if verbose:
            print('hipcc:', ', '.join(os.path.basename(s) for s, _ in jobs))
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr)
        if verbose:
            print('linked', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
