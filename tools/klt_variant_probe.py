#!/usr/bin/env python3
"""Schedules of the one-track KLT kernel against each other: plain (ordered sums as v_add_f64 chains in lanes 0..4), pipe
(SFMX_KLT_PIPE=1) and mfma (SFMX_KLT_SUMS=mfma: the ordered sums as chains of v_mfma_f64_4x4x4 with B = 1.0).
Bit-equality of fwd / back / keep / step counts with the plain kernel and launch duration (HIP events, median of 9)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0)
ctx.set_timing(True)
os.environ["SFMX_KLT_K"] = "0"
VARIANTS = [("plain", {"SFMX_KLT_PIPE": "0", "SFMX_KLT_SUMS": "valu"}), ("pipe", {"SFMX_KLT_PIPE": "1", "SFMX_KLT_SUMS": "valu"}),
            ("mfma", {"SFMX_KLT_PIPE": "0", "SFMX_KLT_SUMS": "mfma"}), ("mfma+pipe", {"SFMX_KLT_PIPE": "1", "SFMX_KLT_SUMS": "mfma"})]
if len(sys.argv) > 1:
    VARIANTS = [v for v in VARIANTS if v[0] in sys.argv[1].split(",") or v[0] == "plain"]


def run(pa, pb, pts, env, radius=5):
    os.environ.update(env)
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
        ref = None
        cells = []
        for name, env in VARIANTS:
            (f, bk, k, s), med, mn = run(pa, pb, pts[:T], env, radius)
            if ref is None:
                ref = (f, bk, k, s)
                same = "ref"
            else:
                same = "identical" if (np.array_equal(ref[0].view(np.uint64), f.view(np.uint64)) and np.array_equal(ref[1].view(np.uint64), bk.view(np.uint64))
                                       and np.array_equal(ref[2], k) and ref[3] == s) else "MISMATCH"
            cells.append(f"{name} {med:7.1f} us (min {mn:7.1f}) {same}")
        print(f"{tag} r={radius} T={T:5d} | " + " | ".join(cells) + f" | lk_steps {ref[3]}", flush=True)


seq = I.synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
pts = I.corners(ctx, ctx.pyramid(seq["images"][0], 3), 2200)
sweep("bench", seq["images"][0], seq["images"][1], pts, (64, 500, 1024, 1240, 1564, len(pts)))
for rad in (1, 2, 3, 4, 6, 7):
    sweep("bench", seq["images"][0], seq["images"][1], pts, (64, 1240, 1564) if rad == 4 else (64, 1240), radius=rad)
edge = np.array([[0.2, 0.3], [639.5, 479.5], [-3.0, 10.0], [5.0, -2.5], [638.9, 100.0], [100.0, 478.7], [1e12, 5.0], [np.nan, 7.0], [320.0, 240.0],
                 [-40.0, -40.0], [700.0, 500.0], [15.5, 15.5], [16.0, 464.0]])
for rad in (5, 2, 7):
    sweep("edge ", seq["images"][0], seq["images"][1], np.ascontiguousarray(edge), (len(edge),), radius=rad)
seq3 = I.synth.make_sequence(2, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
pts3 = I.corners(ctx, ctx.pyramid(seq3["images"][0], 3), 5000, min_dist=4)
sweep("c3   ", seq3["images"][0], seq3["images"][1], pts3, (2500, 5000))
