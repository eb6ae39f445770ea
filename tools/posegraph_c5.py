#!/usr/bin/env python3
"""BASELINE config 5 size: pose graph of N = 10 000 keyframes (3N = 30 000 unknowns).  Times the structured FP64-MFMA solve
(sfmx_posegraph_solve: H = L (x) I_3, blocked Cholesky on the matrix cores) and checks it with an independent residual in
numpy (|L X - G| / |G|) -- the dense reference system would need 7.2 GB and hours on the CPU.  Smaller
sizes are also compared with numpy's solve.  Product path only.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0); ctx.set_timing(True)


def system(N, loops, seed):
    rng = np.random.default_rng(seed)
    edges = [(i, i + 1, 1.0) for i in range(N - 1)]
    for _ in range(loops):
        a = int(rng.integers(0, N - 8))
        edges.append((a, int(rng.integers(a + 6, N)), 2.0))
    ent = {}
    G = np.zeros((N, 3))
    for i, j, w in edges:
        r = rng.normal(size=3) * 0.01
        for (a, b, s) in ((i, i, w), (j, j, w), (max(i, j), min(i, j), -w)):
            ent[(a, b)] = ent.get((a, b), 0.0) + s
        G[i] -= w * r
        G[j] += w * r
    ent[(0, 0)] += 1e9
    G[0] = 0
    ij = np.array(list(ent.keys()), np.int32)
    v = np.array(list(ent.values()))
    return ij, v, G


def residual(ij, v, G, X):
    R = np.zeros_like(X)
    for (a, b), s in zip(ij, v):
        R[a] += s * X[b]
        if a != b:
            R[b] += s * X[a]
    return np.abs(R - G).max() / np.abs(G).max()


for N, loops in ((1000, 25), (3000, 75), (10000, 250)):
    ij, v, G = system(N, loops, 1)
    rc, X = ctx.posegraph_solve(N, ij, v, G)
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); rc, X = ctx.posegraph_solve(N, ij, v, G); t.append(time.perf_counter() - t0)
    line = f"N={N} keyframes (3N={3*N} unknowns), {loops} loop edges: rc={rc}, wall {min(t)*1e3:.1f} ms, kernels {ctx.last_kernel_us()/1e3:.1f} ms, residual {residual(ij, v, G, X):.2e}"
    if N <= 3000:
        L = np.zeros((N, N))
        for (a, b), s in zip(ij, v):
            L[a, b] = s; L[b, a] = s
        Xn = np.linalg.solve(L, G)
        line += f", vs numpy solve {np.abs(X - Xn).max() / np.abs(Xn).max():.2e}"
    print(line, flush=True)
