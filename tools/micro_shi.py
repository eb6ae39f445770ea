import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
capi, synth = I.capi, I.synth
ctx = capi.Context(0)
seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
pa = ctx.pyramid(seq["images"][0], 3); pb = ctx.pyramid(seq["images"][1], 3)
for _ in range(3): r = ctx.shi_candidates_pruned(pa, 0.01, 8); ctx.shi_candidates_pruned(pb, 0.01, 8)
t0 = time.perf_counter()
for _ in range(20): ctx.shi_candidates_pruned(pa, 0.01, 8); ctx.shi_candidates_pruned(pb, 0.01, 8)
print("shi pruned call wall us:", (time.perf_counter() - t0) / 40 * 1e6, "survivors", r[4], "of", r[5], "undecided", int((~r[2]).sum()), "graph", os.environ.get("SFMX_NO_GRAPH") is None)
# interleaved with other API calls, as in the pipeline
pts = I.corners(ctx, pa, 2200)
N = 1100
xi, xj = I.two_view(N)
idx8 = I.octets(N, 2500)
tt = 0.0
for i in range(20):
    ctx.klt_track(pa, pb, pts)
    t0 = time.perf_counter(); ctx.shi_candidates_pruned(pa if i % 2 else pb, 0.01, 8); tt += time.perf_counter() - t0
    ctx.ransac_score(xi, xj, idx8, 1e-3)
print("interleaved shi call wall us:", tt / 20 * 1e6)
