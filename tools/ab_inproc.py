#!/usr/bin/env python3
"""A/B of environment switches INSIDE one process: the variants take turns (rep 0: A B C, rep 1: A B C, ...), a few passes of the
bench workload each, so that drift of the box and of the process hits all of them alike.  Only for switches the library reads
per call (SFMX_KLT_SUMS / _PIPE / _K, SFMX_RANSAC_HYP, ...), not for those read once per process.
usage: tools/ab_inproc.py [--reps 30] [--passes 3] name:K=V[,K=V] name2:K=V ..."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib, torch
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("variants", nargs="+")
args = ap.parse_args()
variants = []
for v in args.variants:
    name, _, kv = v.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(";" if ";" in kv else ",") if x)))  # K=V pairs separated by "," (or ";" when a value holds a comma)
keys = sorted({k for _, e in variants for k in e})
pipe = importlib.import_module(I.PKG + ".pipeline")
seq = I.synth.make_sequence(47, 640, 480, 0.3, n_blobs=20000, seed=7)
cfg = dict(pipe.DEFAULTS, frames=47, max_tracks=2200, min_tracks=900, export_pointcloud=0)
ctx = I.capi.Context(0)
dev = torch.from_numpy(np.ascontiguousarray(seq["images"])).to("cuda:0")
torch.cuda.synchronize()


def one():
    t0 = time.perf_counter()
    r = pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, None, images_dev=dev.data_ptr(), shape=tuple(dev.shape))
    return time.perf_counter() - t0, r


for _ in range(3):
    one()
ms = {n: [] for n, _ in variants}
klt = {n: [] for n, _ in variants}
extra = {n: {k: [] for k in ("sec_lane_b_busy", "sec_join_wait", "sec_m_kf", "sec_feed_wait", "sec_m_step", "sec_m_ransac", "sec_shi_wait")} for n, _ in variants}
ref_log = None
for rep in range(args.reps):
    for name, env in variants:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(args.passes):
            dt, r = one()
            ms[name].append(dt * 1e3)
            klt[name].append(r["stats"]["sec_klt"] * 1e3)
            for k in extra[name]:
                extra[name][k].append(r["stats"][k] * 1e3)
            if ref_log is None:
                ref_log = r["log"]
            assert r["log"] == ref_log, f"variant {name} changed the output"
for name, _ in variants:
    a = np.array(ms[name])
    print(f"{name:14s} passes {len(a):3d}: median {np.median(a):6.2f} ms  mean {a.mean():6.2f}  p10 {np.percentile(a, 10):6.2f}  p90 {np.percentile(a, 90):6.2f}"
          f"  -> {47e3 / np.median(a):7.1f} keyframes/s (median) | tracker-lane KLT wall {np.median(klt[name]):5.2f} ms | "
          + " ".join(f"{k[4:]} {np.median(v):5.2f}" for k, v in extra[name].items()), flush=True)
