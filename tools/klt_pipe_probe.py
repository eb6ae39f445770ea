#!/usr/bin/env python3
"""k_klt_track with the ordered sums of the first 64 pixels scheduled under the second half of the sample grid (SFMX_KLT_PIPE=1)
against the plain schedule: bit-equality of fwd / back / keep / step counts and launch duration (HIP events, median of 9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0)
ctx.set_timing(True)
os.environ["SFMX_KLT_K"] = "0"


def run(pa, pb, pts, pipe, radius=5):
    os.environ["SFMX_KLT_PIPE"] = str(pipe)
    us = []
    for _ in range(9):
        out = ctx.klt_track(pa, pb, pts, radius=radius)
        us.append(ctx.last_kernel_us())
    return out, float(np.median(us)), float(np.min(us))


def sweep(tag, a, b, pts, Ts, radius=5):
    pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
    for T in Ts:
        if T > len(pts):
            continue
        (f0, b0, k0, s0), m0, n0 = run(pa, pb, pts[:T], 0, radius)
        (f1, b1, k1, s1), m1, n1 = run(pa, pb, pts[:T], 1, radius)
        same = np.array_equal(f0.view(np.uint64), f1.view(np.uint64)) and np.array_equal(b0.view(np.uint64), b1.view(np.uint64)) and np.array_equal(k0, k1) and s0 == s1
        print(f"{tag} r={radius} T={T:5d} | plain {m0:7.1f} us (min {n0:7.1f}) | pipe {m1:7.1f} us (min {n1:7.1f}) | {'identical' if same else 'MISMATCH'} | lk_steps {s0}", flush=True)


seq = I.synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
pts = I.corners(ctx, ctx.pyramid(seq["images"][0], 3), 2200)
sweep("bench", seq["images"][0], seq["images"][1], pts, (64, 500, 1024, 1240, 1564, len(pts)))
sweep("bench", seq["images"][0], seq["images"][1], pts, (64, 1240), radius=4)
edge = np.array([[0.2, 0.3], [639.5, 479.5], [-3.0, 10.0], [5.0, -2.5], [638.9, 100.0], [100.0, 478.7], [1e12, 5.0], [np.nan, 7.0], [320.0, 240.0],
                 [-40.0, -40.0], [700.0, 500.0], [15.5, 15.5], [16.0, 464.0]])
sweep("edge ", seq["images"][0], seq["images"][1], np.ascontiguousarray(edge), (len(edge),))
seq3 = I.synth.make_sequence(2, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
pts3 = I.corners(ctx, ctx.pyramid(seq3["images"][0], 3), 5000, min_dist=4)
sweep("c3   ", seq3["images"][0], seq3["images"][1], pts3, (2500, 5000))
