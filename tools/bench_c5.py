#!/usr/bin/env python3
"""BASELINE config C5 on a prefix: 1920x1080 frames, loop closures + pose graph + second BA inside the per-frame loop.
The ring camera goes out and back (0.1 deg per frame, --period frames per leg), so every keyframe of a return leg finds its
twin of the outbound leg again (descriptor search, LK + RANSAC verification, T:1822-1866) and each accepted loop closure
runs posegraph_optimize_centers + a second BA.  Frames are generated with numpy (~0.6 s each), hence a prefix
(--frames, default 60) of the 10 000-frame configuration.  --solver structured forces the FP64-MFMA pose-graph solver the
product uses above 6 400 unknowns (2 134 keyframes); the default follows the product's size rule.  Prints one JSON line."""
import argparse, importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import torch
pipe = importlib.import_module(I.PKG + ".pipeline")
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--period", type=int, default=10)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--max-tracks", type=int, default=2200)
ap.add_argument("--solver", choices=("auto", "dense", "structured"), default="auto")
a = ap.parse_args()
if a.solver != "auto":
    os.environ["SFMX_POSEGRAPH_SOLVER"] = a.solver
tri = [abs((f + a.period) % (2 * a.period) - a.period) for f in range(a.frames)]  # period, period-1, .., 0, 1, .., period, ..
ang = [0.1 * (a.period - t) for t in tri]                                          # 0, .1, .., period*.1, .., 0, ..
t0 = time.time()
seq = I.synth.make_sequence(a.frames, 1920, 1080, 0.1, n_blobs=20000, seed=13, angles=ang)
gen_s = time.time() - t0
cfg = dict(pipe.DEFAULTS, frames=a.frames, max_tracks=a.max_tracks, min_tracks=min(900, a.max_tracks * 9 // 22), kf_min_inliers=100, kf_parallax_px=1.0,
           export_pointcloud=0)
ctx = I.capi.Context(0)
dev = torch.from_numpy(np.ascontiguousarray(seq["images"])).to("cuda:0")
torch.cuda.synchronize()
run = lambda timing=False, out=None: pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, out, images_dev=dev.data_ptr(),
                                              shape=tuple(dev.shape), timing=timing)
first = run()
t0 = time.perf_counter()
same = True
for _ in range(a.passes):
    r = run()
    same = same and r["log"] == first["log"] and np.array_equal(r["centres"].view(np.uint64), first["centres"].view(np.uint64))
dt = (time.perf_counter() - t0) / a.passes
p = run(timing=True)["stats"]
s = r["stats"]
import tempfile
with tempfile.TemporaryDirectory() as td:
    run(out=td)
    edges = open(os.path.join(td, "posegraph_edges.csv")).read().splitlines()[1:]
loops = sum(1 for l in edges if l.endswith(",1"))
print(json.dumps({"workload": f"C5 prefix: {a.frames} frames 1920x1080, out-and-back ring path (period {a.period}), max_tracks {a.max_tracks}, "
                              f"loop closure + pose graph + second BA in the loop; pose-graph solver: {a.solver}",
                  "frames_per_s": round(a.frames / dt, 2), "keyframes_per_s": round(s["n_keyframes"] / dt, 2), "ms_per_frame": round(dt / a.frames * 1e3, 3),
                  "n_keyframes": s["n_keyframes"], "n_edges": s["n_edges"], "loop_closures_accepted": loops, "map_points": s["n_points"],
                  "passes_bit_identical": bool(same), "tracks_per_klt_call": round(s["tracks_in"] / max(1, s["klt_calls"]), 1),
                  "klt_kernel_us_per_call": round(p["us_klt_kernel"] / max(1, p["klt_calls"]), 1),
                  "kernel_us_per_pass": {k: round(v[0], 1) for k, v in sorted(p["kernels"].items(), key=lambda kv: -kv[1][0]) if v[1] > 0},
                  "frame_generation_s": round(gen_s, 1),
                  "host_seconds": {k: round(s[k], 4) for k in ("sec_total", "sec_m_step", "sec_m_ransac", "sec_m_kf", "sec_klt", "sec_ransac", "sec_ba", "sec_shi", "sec_join_wait", "sec_pf_busy", "sec_lane_b_busy", "sec_lane_c_busy", "sec_lane_a_busy", "sec_lane_e_busy", "sec_feed_wait", "sec_shi_wait")}}))
