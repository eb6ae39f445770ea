#!/usr/bin/env python3
"""Fixed small workload for rocprofv3 PMC passes: the hot kernels at bench-like sizes.
Usage (GPU box):  rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/prof_kernels.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
# launch size of the KLT kernel: the bench workload tracks ~1240 points per call on average (BENCH line: tracks_in / klt_calls)
TRACKS = int(os.environ.get("SFMX_PROF_TRACKS", "1240"))
ctx = I.capi.Context(0)
seq = I.synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
a, b = seq["images"]
pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
pts = I.corners(ctx, pa, 2200)[:TRACKS]
for _ in range(5):
    fwd, back, keep, steps = ctx.klt_track(pa, pb, pts)
print("klt tracks", len(pts), "lk_steps", steps)
meta = {"klt_tracks": len(pts), "klt_lk_steps": int(steps), "ransac_points": 1100, "ransac_hypotheses": 2500, "ba_W": 6, "ba_P": 600}
if os.environ.get("SFMX_PROF_META"):
    json.dump(meta, open(os.environ["SFMX_PROF_META"], "w"))
N = 1100
xi, xj = I.two_view(N)
idx8 = I.octets(N, 2500)
for _ in range(5):
    ctx.ransac_score(xi, xj, idx8, 1e-3)
print("ransac N", N, "H 2500")
for _ in range(3):
    ctx.shi_candidates_pruned(pa, 0.01, 8)
pw, K, X, ptr, li, uv = I.ba_problem(6, 600)
prob = ctx.ba_problem(6, X, ptr, li, uv)
for _ in range(5):
    prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
print("ba W 6 P 600")
prob.close()
if os.environ.get("SFMX_PROF_C4"):  # BASELINE config C4: W = 10, P = 50 000 (what the chunked row ring moves per iteration)
    pw, K, X, ptr, li, uv = I.ba_problem(10, 50000)
    prob = ctx.ba_problem(10, X, ptr, li, uv)
    for _ in range(3):
        prob.step_sharded_elements(None, pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    print("ba W 10 P 50000")
    prob.close()
ctx.close()
