#!/usr/bin/env python3
"""BASELINE config C3: 640x480, 5 k tracks per frame, KLT/RANSAC throughput on one GPU.
The frames (frame-filling texture, sub-pixel flow -- see tests/test_gpu_pipeline.py for why) are generated with
numpy at ~0.3 s each, so the default run uses a 120-frame prefix of the 1000-frame config; --frames 1000 is the
full one.  Prints one JSON line."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib, torch
pipe = importlib.import_module(I.PKG + ".pipeline")
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--passes", type=int, default=3)
a = ap.parse_args()
t0 = time.time()
seq = I.synth.make_sequence(a.frames, 640, 480, 0.01, n_blobs=150000, seed=7, shell_scale=3.5)
gen_s = time.time() - t0
cfg = dict(pipe.DEFAULTS, frames=a.frames, max_tracks=5000, min_tracks=2045, min_distance=4, kf_parallax_px=1.0, export_pointcloud=0)
ctx = I.capi.Context(0)
dev = torch.from_numpy(np.ascontiguousarray(seq["images"])).to("cuda:0")
torch.cuda.synchronize()
run = lambda timing=False: pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, None, images_dev=dev.data_ptr(),
                                    shape=tuple(dev.shape), timing=timing)
run()
t0 = time.perf_counter()
for _ in range(a.passes):
    r = run()
dt = (time.perf_counter() - t0) / a.passes
p = run(timing=True)["stats"]
s = r["stats"]
tracks = s["tracks_in"] / max(1, s["klt_calls"])
klt_us = p["us_klt_kernel"] / max(1, p["klt_calls"])
print(json.dumps({"workload": f"C3 prefix: {a.frames} frames 640x480, max_tracks 5000, min_distance 4", "frames_per_s": round(a.frames / dt, 2),
                  "keyframes_per_s": round(s["n_keyframes"] / dt, 2), "ms_per_frame": round(dt / a.frames * 1e3, 3),
                  "tracks_per_klt_call": round(tracks, 1), "klt_kernel_us_per_call": round(klt_us, 1),
                  "klt_fp64_tflops": round(12.7e3 * p["lk_steps"] / max(1, p["klt_calls"]) / (klt_us * 1e-6) / 1e12, 3),
                  "ransac_points_per_call": round(s["ransac_points"] / max(1, s["ransac_calls"]), 1), "map_points": s["n_points"],
                  "n_keyframes": s["n_keyframes"], "frame_generation_s": round(gen_s, 1),
                  "host_seconds": {k: round(s[k], 4) for k in ("sec_total", "sec_m_step", "sec_m_ransac", "sec_m_kf", "sec_klt", "sec_ransac", "sec_ba", "sec_shi", "sec_join_wait", "sec_pf_busy", "sec_lane_b_busy", "sec_lane_c_busy", "sec_lane_a_busy", "sec_lane_e_busy", "sec_bookkeeping", "sec_tri_iter", "sec_tri_solve", "sec_tri_insert", "sec_ba_gather", "sec_desc", "sec_feed_wait", "sec_shi_wait")}}))
