import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
PKG = "structure-from-motion-3d-reconstruction_amd"
capi = importlib.import_module(PKG + ".capi"); synth = importlib.import_module(PKG + ".synth")
ctx = capi.Context(0)
seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
pa = ctx.pyramid(seq["images"][0], 3); pb = ctx.pyramid(seq["images"][1], 3)
for _ in range(3): r = ctx.shi_candidates_pruned(pa, 0.01, 8); ctx.shi_candidates_pruned(pb, 0.01, 8)
t0 = time.perf_counter()
for _ in range(20): ctx.shi_candidates_pruned(pa, 0.01, 8); ctx.shi_candidates_pruned(pb, 0.01, 8)
print("shi pruned call wall us:", (time.perf_counter() - t0) / 40 * 1e6, "survivors", r[4], "of", r[5], "undecided", int((~r[2]).sum()), "graph", os.environ.get("SFMX_NO_GRAPH") is None)
# interleaved with other API calls, as in the pipeline
import helpers as H
O = H.oracle()
pts = H.shi_tomasi(O, "orc", seq["images"][0], 2200, 0.01, 8)
g = np.load(os.path.join(H.GOLDEN, "hotpath.npz"))
N = 1100
xi = np.tile(g["tv_xi"], (N // 240 + 1, 1))[:N]; xj = np.tile(g["tv_xj"], (N // 240 + 1, 1))[:N]
idx8 = H.uniform_draws(O, "orc", 12345, N, 8 * 2500).reshape(2500, 8)
tt = 0.0
for i in range(20):
    ctx.klt_track(pa, pb, pts)
    t0 = time.perf_counter(); ctx.shi_candidates_pruned(pa if i % 2 else pb, 0.01, 8); tt += time.perf_counter() - t0
    ctx.ransac_score(xi, xj, idx8, 1e-3)
print("interleaved shi call wall us:", tt / 20 * 1e6)
