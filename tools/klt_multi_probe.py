#!/usr/bin/env python3
"""k_klt_track_multi (K tracks per wavefront) against the one-track kernel: bit-equality of fwd / back / keep / step counts and
launch duration (HIP events inside the library, median of 7) for K = 0 (one-track kernel, f32 windows), 1, 2, 4 at several
track counts, on a bench-like frame pair (T <= 2200) and on a C3-like one (5 000 tracks)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
ctx = I.capi.Context(0)
ctx.set_timing(True)


def run(pa, pb, pts, K):
    os.environ["SFMX_KLT_K"] = str(K)
    us = []
    for _ in range(7):
        out = ctx.klt_track(pa, pb, pts)
        us.append(ctx.last_kernel_us())
    return out, float(np.median(us)), float(np.min(us))


def sweep(tag, a, b, pts, Ts):
    pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
    for T in Ts:
        if T > len(pts):
            continue
        ref = None
        line = [f"{tag} T={T:5d}"]
        for K in (0, 1, 2, 4):
            (fwd, back, keep, steps), med, mn = run(pa, pb, pts[:T], K)
            if ref is None:
                ref = (fwd, back, keep, steps)
                same = True
            else:
                same = (np.array_equal(fwd.view(np.uint64), ref[0].view(np.uint64)) and np.array_equal(back.view(np.uint64), ref[1].view(np.uint64))
                        and np.array_equal(keep, ref[2]) and steps == ref[3])
            line.append(f"K={K}: {med:7.1f} us (min {mn:7.1f}) {'ok ' if same else 'MISMATCH'} slow={ctx.klt_slow_steps()}")
        print(" | ".join(line) + f" | lk_steps {ref[3]}", flush=True)


seq = I.synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
pts = I.corners(ctx, ctx.pyramid(seq["images"][0], 3), 2200)
sweep("bench", seq["images"][0], seq["images"][1], pts, (64, 500, 1024, 1240, 1564, len(pts)))
# tracks near / across the image border and NaN / huge coordinates: the zero-fill and "touches nothing" paths
edge = np.array([[0.2, 0.3], [639.5, 479.5], [-3.0, 10.0], [5.0, -2.5], [638.9, 100.0], [100.0, 478.7], [1e12, 5.0], [np.nan, 7.0], [320.0, 240.0],
                 [-40.0, -40.0], [700.0, 500.0], [15.5, 15.5], [16.0, 464.0]])
sweep("edge ", seq["images"][0], seq["images"][1], np.ascontiguousarray(edge), (len(edge),))
seq3 = I.synth.make_sequence(2, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
pts3 = I.corners(ctx, ctx.pyramid(seq3["images"][0], 3), 5000, min_dist=4)
sweep("c3   ", seq3["images"][0], seq3["images"][1], pts3, (2500, 5000, len(pts3)))
# other window radii (every instantiation) on a few hundred tracks
for r in (1, 2, 3, 4, 6):
    pa, pb = ctx.pyramid(seq["images"][0], 3), ctx.pyramid(seq["images"][1], 3)
    ref = None
    oks = []
    for K in (0, 1, 2, 4):
        os.environ["SFMX_KLT_K"] = str(K)
        out = ctx.klt_track(pa, pb, pts[:300], radius=r)
        if ref is None:
            ref = out
        oks.append(all(np.array_equal(np.asarray(x).view(np.uint8), np.asarray(y).view(np.uint8)) for x, y in zip(out[:3], ref[:3])) and out[3] == ref[3])
    print(f"radius {r}: K=0,1,2,4 identical: {oks}", flush=True)
