#!/usr/bin/env python3
"""GEMM engine experiments, interleaved in ONE process (rule: perf deltas come
from interleaved rounds on one device). Each config line:
   name n ta tb flags order swizzle tile waves same_ab
usage: gemm_exp.py rounds < configs.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib

def units(n, flags, tile=128):
    T = n // 128
    u = 0
    for i in range(T):
        for j in range(T):
            if (flags & 1) and j < i:
                continue
            lo, hi = 0, T
            if flags & 2: lo = max(lo, i)
            if flags & 4: hi = min(hi, i + 1)
            if flags & 8: lo = max(lo, j)
            if flags & 16: hi = min(hi, j + 1)
            u += max(0, hi - lo)
    return u

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfgs = []
for line in sys.stdin:
    p = line.split()
    if not p or p[0].startswith('#'):
        continue
    cfgs.append((p[0],) + tuple(int(x) for x in p[1:]))
dev = _lib.Handle(0)
res = {c[0]: [] for c in cfgs}
for r in range(rounds + 1):
    for c in cfgs:
        name, n, ta, tb, flags, order, swz, tile, waves, same = c
        ms = dev.la_gemm_bench_ex(n, ta, tb, flags, order, swz, tile, waves, same, reps=1)
        if r > 0:
            res[name].append(ms)
for c in cfgs:
    name, n, ta, tb, flags = c[:5]
    fl = units(n, flags) * 2.0 * 128 ** 3
    t = np.array(res[name])
    print('%-28s n=%5d ta=%d tb=%d flags=%2d: median %8.3f ms  min %8.3f  -> %6.2f TF (median)'
          % (name, n, ta, tb, flags, np.median(t), t.min(), fl / np.median(t) * 1e-9))
