import os, sys
sys.path.insert(0, "tools")
import _inputs as I
ctx = I.capi.Context(0); ctx.set_timing(True)
seq = I.synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
a, b = seq["images"]
pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
pts = I.corners(ctx, pa, 2200)
for T in (64, 1564):
    for _ in range(2):
        ctx.klt_track(pa, pb, pts[:T])
        print("T", T, "us", ctx.last_kernel_us(), flush=True)
