#!/usr/bin/env python3
"""Where do the two BA window kernels lose their time inside the pipeline?  Needs the diagnostic build
   make -C <pkg>/csrc OUT=<pkg>/_build_diag HIPFLAGS_EXTRA=-DSFMX_BA_WGSTAMPS hip host
in which every workgroup of k_ba_points_window / k_ba_reduce records {start, end (100 MHz), hardware id, block}.  Prints, per
kernel, for launches with the device to itself and for launches inside the pipeline: span of the launch (first start -> last
end), skew of the workgroup starts, workgroup durations, workgroups per CU."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib, torch
diag = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), I.PKG, "_build_diag")
I.capi.LIB_PATH = os.path.join(diag, "libsfmx.so")
pipe = importlib.import_module(I.PKG + ".pipeline")
pipe.HOST_LIB_PATH = os.path.join(diag, "libsfmx_host.so")
lib = I.capi.load_library()
ctx = I.capi.Context(0)
TMP = "/tmp/wgstamps.bin"


def dump():
    n = lib.sfmx_debug_dump_wgstamps(TMP.encode())
    a = np.fromfile(TMP, dtype=np.uint64).reshape(-1, 8) if n > 0 else np.zeros((0, 8), np.uint64)
    return a


def launches(a, kind):
    m = (a[:, 3] >> np.uint64(48)) == kind
    r = a[m]
    r = r[np.argsort(r[:, 0])]
    out, cur, seen = [], [], set()
    for row in r:
        blk = int((row[3] >> np.uint64(24)) & np.uint64(0xffffff))
        nblk = int(row[3] & np.uint64(0xffffff))
        if blk in seen or (cur and int(row[0]) - int(cur[0][0]) > 100000):  # same block again / 1 ms later: next launch
            out.append(np.array(cur)); cur, seen = [], set()
        cur.append(row); seen.add(blk)
        if len(cur) == nblk:
            out.append(np.array(cur)); cur, seen = [], set()
    return [l for l in out if len(l) == int(l[0][3] & np.uint64(0xffffff))]


def report(tag, a):
    for kind, name in ((1, "k_ba_points_window"), (2, "k_ba_reduce")):
        ls = launches(a, kind)
        if not ls:
            print(f"{tag} {name}: no complete launches"); continue
        span = np.array([(l[:, 1].max() - l[:, 0].min()) / 100.0 for l in ls])          # us
        skew = np.array([(l[:, 0].max() - l[:, 0].min()) / 100.0 for l in ls])
        dur_med = np.array([np.median(l[:, 1] - l[:, 0]) / 100.0 for l in ls])
        dur_max = np.array([(l[:, 1] - l[:, 0]).max() / 100.0 for l in ls])
        # CU identity: XCC (bits 28..31), SE (13..15), SH (12), CU (8..11)
        def cus(l):
            hw = l[:, 2].astype(np.uint64)
            key = ((hw >> np.uint64(28)) & np.uint64(15)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(255))
            u, c = np.unique(key, return_counts=True)
            return len(u), c.max()
        ncu = np.array([cus(l)[0] for l in ls]); percu = np.array([cus(l)[1] for l in ls])
        q = lambda v: f"{np.median(v):6.1f} (p10 {np.percentile(v,10):5.1f}, p90 {np.percentile(v,90):6.1f})"
        print(f"{tag} {name}: {len(ls)} launches x {len(ls[0])} workgroups | span us {q(span)} | start skew us {q(skew)} | workgroup us median {q(dur_med)} "
              f"| slowest workgroup us {q(dur_max)} | CUs used {np.median(ncu):.0f}, most workgroups on one CU {np.median(percu):.0f}", flush=True)
        # phase marks (us after the workgroup's own start; median over workgroups and launches; 0 = mark not reached)
        allr = np.concatenate(ls)
        marks = []
        for i in range(4):
            m = allr[:, 4 + i].astype(np.int64) - allr[:, 0].astype(np.int64)
            m = m[allr[:, 4 + i] != 0]
            marks.append(f"m{i} {np.median(m) / 100.0:5.1f}" if len(m) else f"m{i}   -  ")
        end = (allr[:, 1].astype(np.int64) - allr[:, 0].astype(np.int64)) / 100.0
        names = "inputs | observations | sums+inverse+gain | rows issued" if kind == 1 else "first loads issued | chains done | ticket taken | -"
        print(f"{tag} {name}: marks ({names}): " + "  ".join(marks) + f"  end {np.median(end):5.1f}", flush=True)


# ---- alone
pw, K, X, ptr, li, uv = I.ba_problem(6, 600)
prob = ctx.ba_problem(6, X, ptr, li, uv)
for _ in range(3):
    prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
dump()
for _ in range(40):
    prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
report("alone   ", dump())
prob.close()
if os.environ.get("STAMPS_ALONE_ONLY"):
    sys.exit(0)
# ---- inside the pipeline
seq = I.synth.make_sequence(47, 640, 480, 0.3, n_blobs=20000, seed=7)
cfg = dict(pipe.DEFAULTS, frames=47, max_tracks=2200, min_tracks=900, export_pointcloud=0)
dev = torch.from_numpy(np.ascontiguousarray(seq["images"])).to("cuda:0")
torch.cuda.synchronize()
run = lambda: pipe.run(ctx, None, seq["names"], seq["K"], seq["lat"], seq["lon"], cfg, None, images_dev=dev.data_ptr(), shape=tuple(dev.shape))
for _ in range(3):
    run()
dump()
for _ in range(4):
    run()
report("pipeline", dump())
