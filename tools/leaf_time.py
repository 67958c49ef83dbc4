#!/usr/bin/env python3
"""time the leaf kernel inside a potrf run: python tools/leaf_time.py (uses GPX_LEAF_SKIP)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
dev = _lib.Handle(0)
try:
    ms = dev.la_potrf_bench(4096, True, reps=3)
    print('skip=%s potrf+inv n=4096: %.3f ms' % (os.environ.get('GPX_LEAF_SKIP', '0'), ms))
except Exception as e:
    print('skip=%s failed: %s' % (os.environ.get('GPX_LEAF_SKIP', '0'), str(e)[:80]))
