#!/usr/bin/env python3
"""Per-pass wall time of the bench workload, outside (Python) and inside (sfmx_pipeline_run) the C call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib, torch
pipe = importlib.import_module(I.PKG + ".pipeline")
seq = I.synth.make_sequence(47, 640, 480, 0.3, n_blobs=20000, seed=7)
cfg = dict(pipe.DEFAULTS, frames=47, max_tracks=2200, min_tracks=900, export_pointcloud=0)
ctx = I.capi.Context(0)
dev = torch.from_numpy(np.ascontiguousarray(seq["images"])).to("cuda:0")
torch.cuda.synchronize()
for i in range(8):
    t0 = time.perf_counter()
    r = pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, None, images_dev=dev.data_ptr(), shape=tuple(dev.shape))
    dt = time.perf_counter() - t0
    s = r["stats"]
    print(f"pass {i}: python {dt*1e3:.2f} ms, sec_wall {s['sec_wall']*1e3:.2f}, sec_total {s['sec_total']*1e3:.2f}, setup {s['sec_setup']*1e3:.2f}")
