#!/usr/bin/env python3
"""Per-kernel GPU timings (HIP events inside libsfmx) at bench-like sizes.  Run on the GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
PKG = "structure-from-motion-3d-reconstruction_amd"
capi = importlib.import_module(PKG + ".capi"); synth = importlib.import_module(PKG + ".synth")
import helpers as H
ctx = capi.Context(0); ctx.set_timing(True)
def med(f, n=7):
    v = []
    for _ in range(n):
        f(); v.append(ctx.last_kernel_us())
    return float(np.median(v)), float(np.min(v))
rng = np.random.default_rng(0)
for n in (12, 36, 60, 141):
    M = rng.normal(size=(n, n)); A = M @ M.T + np.eye(n) * 1e-3; b = rng.normal(size=n)
    print(f"solve n={n}: med/min us", med(lambda: ctx.solve_dense(A, b)))
seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
a, b_ = seq["images"]
O = H.oracle()
pts = H.shi_tomasi(O, "orc", a, 2200, 0.01, 8)
pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b_, 3)
for T in (len(pts), 500, 64):
    r = med(lambda: ctx.klt_track(pa, pb, pts[:T]))
    fwd, back, keep, steps = ctx.klt_track(pa, pb, pts[:T])
    print(f"klt T={T}: med/min us {r}, lk_steps {steps}, kept {int(keep.sum())}")
print("shi score:", med(lambda: ctx.shi_score(pa)))
print("shi cand pruned:", med(lambda: ctx.shi_candidates_pruned(pa, 0.01, 8)), "n,ntot=", ctx.shi_candidates_pruned(pa, 0.01, 8)[4:6])
g = np.load(os.path.join(H.GOLDEN, "hotpath.npz"))
for N in (240, 1500):
    xi = np.tile(g["tv_xi"], (N // 240 + 1, 1))[:N] + rng.normal(size=(N, 2)) * 1e-4 * (N != 240)
    xj = np.tile(g["tv_xj"], (N // 240 + 1, 1))[:N] + rng.normal(size=(N, 2)) * 1e-4 * (N != 240)
    idx8 = H.uniform_draws(O, "orc", 12345, N, 8 * 2500).reshape(2500, 8)
    print(f"ransac N={N} H=2500 (hyp+score+argmax):", med(lambda: ctx.ransac_score(xi, xj, idx8, 1e-3)))
W, P = 6, 600
pw = np.zeros((W, 12))
for k in range(W):
    R, t = synth.ring_pose(2.0 * k); pw[k, :9], pw[k, 9:] = R.ravel(), t
K = synth.K_TEMPLE
X = rng.normal(size=(P, 3)) * 0.05
ptr = np.arange(0, (P + 1) * W, W, dtype=np.int32); li = np.tile(np.arange(W, dtype=np.int32), P)
uv = np.zeros((P * W, 2))
for p in range(P):
    for k in range(W):
        Xc = pw[k, :9].reshape(3, 3) @ X[p] + pw[k, 9:]
        uv[p * W + k] = [K[0, 0] * Xc[0] / Xc[2] + K[0, 2] + rng.normal(), K[1, 1] * Xc[1] / Xc[2] + K[1, 2] + rng.normal()]
prob = ctx.ba_problem(W, X, ptr, li, uv)
print("ba build (points+expand+reduce) W=6 P=600:", med(lambda: prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)))
t0 = time.perf_counter()
for _ in range(50): prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
print("ba step wall us (build+solve+D2H):", (time.perf_counter() - t0) / 50 * 1e6)
