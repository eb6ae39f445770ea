#!/usr/bin/env python3
"""Dense solves at pose-graph sizes, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_solve.py`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
M = rng.normal(size=(n, n)); A = M @ M.T + np.eye(n) * 1e-3; b = rng.normal(size=n)
for _ in range(5):
    ctx.solve_dense(A, b)
ctx.close()
