#!/usr/bin/env python3
"""k_hypotheses with this round's rotation loop (lean) against the loop as of round 2 (SFMX_RANSAC_HYP=legacy): bit-equality of
E, conditioning estimate, flags and counts on bench-like correspondences, kernel time of a call (hypotheses + scoring, HIP
events) and -- with SFMX_RANSAC_TICKS=1 in the environment -- s_memtime ticks per rotation on stderr.  Product path only."""
import ctypes, os, sys
os.environ.setdefault("SFMX_RANSAC_MIN_COND", "0")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib
pipe = importlib.import_module(I.PKG + ".pipeline")
ctx = I.capi.Context(0)
ctx.set_timing(True)
host = pipe.load_host_library()
seq = I.synth.make_sequence(3, 640, 480, 0.3, n_blobs=20000, seed=7)
T = pipe.Tracker(ctx, 640, 480)
for f in range(3):
    prev, cur, ids = T.step(seq["images"][f])
Kinv = np.linalg.inv(seq["K"])


def norm(p):
    h = np.c_[p, np.ones(len(p))] @ Kinv.T
    return np.ascontiguousarray(h[:, :2] / h[:, 2:3])


xi, xj = norm(prev), norm(cur)
rng = np.random.default_rng(5)
cases = [("tracker", xi, xj), ("tracker-240", xi[:240], xj[:240]),
         ("noisy", xi + rng.normal(0, 2e-3, xi.shape), xj + rng.normal(0, 2e-3, xj.shape))]
H = 2500
for tag, a, b in cases:
    n = len(a)
    idx8 = np.zeros(8 * H, np.int32)
    host.sfmx_host_uniform_draws(ctypes.c_uint(12345), ctypes.c_int(n), ctypes.c_int(8 * H), idx8.ctypes.data_as(ctypes.c_void_p))
    idx8 = idx8.reshape(H, 8)
    out = {}
    for var in ("legacy", "lean"):
        os.environ["SFMX_RANSAC_HYP"] = var
        us = []
        for _ in range(7):
            res = ctx.ransac_score_ex(a, b, idx8, 1e-3)
            us.append(ctx.last_kernel_us())
        out[var] = (res, float(np.median(us)))
    r0, r1 = out["legacy"][0], out["lean"][0]
    same = all(np.array_equal(np.ascontiguousarray(r0[k]).view(np.uint8), np.ascontiguousarray(r1[k]).view(np.uint8)) for k in ("E", "cond", "flags", "counts", "lo", "hi"))
    same = same and r0["best_iter"] == r1["best_iter"] and r0["best_count"] == r1["best_count"]
    print(f"{tag:12s} n={n:5d} H={H}: legacy {out['legacy'][1]:7.1f} us | lean {out['lean'][1]:7.1f} us | {'identical' if same else 'MISMATCH'} | best {r1['best_iter']} / {r1['best_count']}", flush=True)
T.close(); ctx.close()
