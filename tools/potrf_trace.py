#!/usr/bin/env python3
"""Target of rocprofv3 --kernel-trace: a few value-only factorisations (or full
evaluations) at size N. usage: potrf_trace.py N reps [inverse]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygp_amd import _lib
n = int(sys.argv[1]); reps = int(sys.argv[2]); inv = len(sys.argv) > 3
dev = _lib.Handle(0)
print(dev.la_potrf_bench(n, inv, reps=reps))
