#!/usr/bin/env python3
"""Does the shader clock decide the duration of the lone-wave kernels?  Times k_solve_regs<36> and the KLT launch
(1) on an otherwise idle device, (2) while a side stream keeps the device busy with large matrix products, and samples
the shader clock (sysfs pp_dpm_sclk / rocm-smi) in both states.  Run on the GPU box."""
import glob, os, subprocess, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import torch
capi, synth = I.capi, I.synth


def sclk():
    out = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for ln in open(p).read().splitlines():
                if ln.strip().endswith("*"):
                    out.append(ln.strip())
        except OSError as e:
            out.append(f"{p}: {e}")
    if not out:
        try:
            r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20)
            out = [ln for ln in r.stdout.splitlines() if "sclk" in ln]
        except Exception as e:  # noqa: BLE001
            out = [repr(e)]
    return out


ctx = capi.Context(0); ctx.set_timing(True)
rng = np.random.default_rng(0)
n = 36
M = rng.normal(size=(n, n)); A = M @ M.T + np.eye(n) * 1e-3; b = rng.normal(size=n)
seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
a, b_ = seq["images"]
pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b_, 3)
pts = I.corners(ctx, pa, 2200)


def measure(tag):
    v = []
    for _ in range(40):
        ctx.solve_dense(A, b); v.append(ctx.last_kernel_us())
    k = []
    for _ in range(10):
        ctx.klt_track(pa, pb, pts[:1240]); k.append(ctx.last_kernel_us())
    print(f"{tag}: solve36 med/min us {np.median(v):.1f}/{np.min(v):.1f}   klt T=1240 med/min us {np.median(k):.1f}/{np.min(k):.1f}   sclk {sclk()}", flush=True)


print("idle sclk", sclk(), flush=True)
measure("alone")
stop = False


def burner(size, sleep):
    s = torch.cuda.Stream()
    x = torch.randn(size, size, device="cuda", dtype=torch.float32)
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(4):
                y = x @ x
            s.synchronize()
            if sleep:
                time.sleep(sleep)


for size, sleep in ((4096, 0.0), (1024, 0.0), (256, 0.0)):
    stop = False
    th = threading.Thread(target=burner, args=(size, sleep)); th.start()
    time.sleep(1.0)
    measure(f"with burner {size}^3 fp32")
    stop = True; th.join()
time.sleep(1.0)
measure("alone again")
