"""SFMX_SOLVE_STAMPS=1 python tools/solve_stamps.py: cycle stamps inside k_solve_regs<36> (stderr of libsfmx)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0)
rng = np.random.default_rng(0)
n = 36
M = rng.normal(size=(n, n)); A = M @ M.T + np.eye(n) * 1e-3; b = rng.normal(size=n)
for _ in range(4):
    ctx.solve_dense(A, b)
