#!/usr/bin/env python3
"""Per-kernel GPU timings (HIP events inside libsfmx) at bench-like sizes.  Run on the GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
capi, synth = I.capi, I.synth
ctx = capi.Context(0); ctx.set_timing(True)
def med(f, n=7):
    v = []
    for _ in range(n):
        f(); v.append(ctx.last_kernel_us())
    return float(np.median(v)), float(np.min(v))
rng = np.random.default_rng(0)
for n in (12, 36, 60, 141, 300, 600):
    M = rng.normal(size=(n, n)); A = M @ M.T + np.eye(n) * 1e-3; b = rng.normal(size=n)
    print(f"solve n={n}: med/min us", med(lambda: ctx.solve_dense(A, b)))
seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
a, b_ = seq["images"]
pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b_, 3)
pts = I.corners(ctx, pa, 2200)
for T in (len(pts), 500, 64):
    r = med(lambda: ctx.klt_track(pa, pb, pts[:T]))
    fwd, back, keep, steps = ctx.klt_track(pa, pb, pts[:T])
    print(f"klt T={T}: med/min us {r}, lk_steps {steps}, per-pixel-path steps {ctx.klt_slow_steps()}, kept {int(keep.sum())}")
print("shi score:", med(lambda: ctx.shi_score(pa)))
print("shi cand pruned:", med(lambda: ctx.shi_candidates_pruned(pa, 0.01, 8)), "n,ntot=", ctx.shi_candidates_pruned(pa, 0.01, 8)[4:6])
for N in (240, 1500, 5000):
    xi, xj = I.two_view(N)
    idx8 = I.octets(N, 2500)
    print(f"ransac N={N} H=2500 (hyp+score+argmax):", med(lambda: ctx.ransac_score(xi, xj, idx8, 1e-3)))
W, P = 6, 600
pw, K, X, ptr, li, uv = I.ba_problem(W, P)
prob = ctx.ba_problem(W, X, ptr, li, uv)
print("ba build (points+expand+reduce) W=6 P=600:", med(lambda: prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)))
t0 = time.perf_counter()
for _ in range(50): prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
print("ba step wall us (event-timed path: build+solve+D2H+stream sync):", (time.perf_counter() - t0) / 50 * 1e6)
ctx.set_timing(False)  # the path the pipeline takes: result polled from pinned memory, window system solved on the host core
for _ in range(20): prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
t0 = time.perf_counter()
for _ in range(200): prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
print("ba step wall us (polled S|b + host solve, through ctypes):", (time.perf_counter() - t0) / 200 * 1e6)
ctx.set_timing(True)

# BASELINE config C4: W=10, P=50 000, every point seen in every pose
W, P = 10, 50000
pw, K, X, ptr, li, uv = I.ba_problem(W, P)
prob4 = ctx.ba_problem(W, X, ptr, li, uv)
print("ba build C4 (W=10 P=50000, 500k residuals) kernel us:", med(lambda: prob4.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3), 5))
t0 = time.perf_counter()
for _ in range(5): prob4.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
print("ba step C4 wall us (build+solve+D2H):", (time.perf_counter() - t0) / 5 * 1e6)
