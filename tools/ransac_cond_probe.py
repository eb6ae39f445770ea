#!/usr/bin/env python3
"""Distribution of the conditioning estimate of the device hypotheses on bench-like frame-to-frame correspondences, how
far the device E is from the exact host E as a function of it, how many hypotheses each exactness rule sends to the host,
and what a call costs.  PROBE_FRAMES=n walks the first n frames of the bench sequence (default 3: one pair in detail;
47: every frame pair of the bench workload, one summary line per pair and the worst dE * cond at the end).
Product path only (no oracle).  Run on the GPU box."""
import ctypes, os, sys, time
os.environ.setdefault("SFMX_RANSAC_MIN_COND", "0")  # keep every non-repeated hypothesis on the device for this probe
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _inputs as I
import importlib
pipe = importlib.import_module(I.PKG + ".pipeline")
ctx = I.capi.Context(0)
host = pipe.load_host_library()
deg = float(os.environ.get("PROBE_DEG", "0.3"))
nframes = int(os.environ.get("PROBE_FRAMES", "3"))
seq = I.synth.make_sequence(nframes, 640, 480, deg, n_blobs=20000, seed=7)
T = pipe.Tracker(ctx, 640, 480)
K = seq["K"]
Kinv = np.linalg.inv(K)
H = 2500


def norm(p):
    h = np.c_[p, np.ones(len(p))] @ Kinv.T
    return np.ascontiguousarray(h[:, :2] / h[:, 2:3])


def probe(xi, xj, seed, detail):
    n = len(xi)
    idx8 = np.zeros(8 * H, np.int32)
    host.sfmx_host_uniform_draws(ctypes.c_uint(seed), ctypes.c_int(n), ctypes.c_int(8 * H), idx8.ctypes.data_as(ctypes.c_void_p))
    idx8 = idx8.reshape(H, 8)
    res = ctx.ransac_score_ex(xi, xj, idx8, 1e-3)
    dt = 0.0
    if detail:
        t0 = time.perf_counter()
        for _ in range(10):
            res = ctx.ransac_score_ex(xi, xj, idx8, 1e-3)
        dt = (time.perf_counter() - t0) / 10
    rep = np.array([len(set(r)) < 8 for r in idx8])
    ex = res["flags"].astype(bool)
    cond = res["cond"]
    E = res["E"]
    Eex = np.zeros((H, 3, 3))
    nz = ~ex
    for h in np.nonzero(nz)[0]:
        o = np.ascontiguousarray(idx8[h], np.int32)
        host.sfmx_host_eight_point_E(xi.ctypes.data_as(ctypes.c_void_p), xj.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p),
                                     Eex[h].ctypes.data_as(ctypes.c_void_p))
    d = np.abs(E - Eex).max(axis=(1, 2)) / np.maximum(np.abs(Eex).max(axis=(1, 2)), 1e-300)
    cex = np.zeros(H, np.int32)
    for h in np.nonzero(nz)[0]:
        m, c = ctx.sampson_mask(xi, xj, Eex[h], 1e-3)
        cex[h] = c
    bad = nz & (res["counts"] != cex)
    out = nz & ((cex < res["lo"]) | (cex > res["hi"]))
    worst = float((d[nz] * cond[nz]).max()) if nz.any() else 0.0
    if detail:
        print(f"n={n} H={H}: repeated-index {rep.sum()}, exact total {ex.sum()}, call {dt*1e3:.2f} ms")
        print("cond percentiles (non-repeated):", np.percentile(cond[~rep], [0, 1, 5, 25, 50, 75, 95, 100]))
        for thr in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
            print(f"  cond < {thr:g}: {(cond[~rep] < thr).sum()}")
        print("rel |E_dev - E_exact| over device rows: max %.3g, median %.3g" % (d[nz].max(), np.median(d[nz])))
        for lo, hi in ((0, 1e-10), (1e-10, 1e-9), (1e-9, 1e-8), (1e-8, 1e-7), (1e-7, 1e-6), (1e-6, 1e-5), (1e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 10)):
            m = nz & (cond >= lo) & (cond < hi)
            if m.any():
                print(f"  cond in [{lo:g},{hi:g}): {m.sum():5d} rows, max rel dE {d[m].max():.3g}, max dE*cond {(d[m]*cond[m]).max():.3g}")
        print("uncertain (lo<hi):", int((res["lo"] < res["hi"]).sum()), " band width counts:", np.bincount(res["hi"] - res["lo"])[:6])
        print("device rows whose count differs from the exact one:", int(bad.sum()), " outside [lo,hi]:", int(out.sum()), " max |diff|:",
              int(np.abs(res["counts"] - cex)[nz].max()))
    return dict(n=n, exact=int(ex.sum()), worst=worst, differ=int(bad.sum()), outside=int(out.sum()), uncertain=int((res["lo"] < res["hi"]).sum()))


rows = []
for f in range(nframes):
    prev, cur, ids = T.step(seq["images"][f])
    if f == 0 or len(prev) < 16:
        continue
    detail = (f == 2) or nframes <= 3 and f == nframes - 1
    if nframes > 3 or detail:
        r = probe(norm(prev), norm(cur), 12345 + f, detail)
        rows.append(r)
        if nframes > 3:
            print(f"frame {f-1:2d}->{f:2d}: n={r['n']:5d} host-exact {r['exact']:4d} uncertain {r['uncertain']:3d} worst dE*cond {r['worst']:.3g} "
                  f"counts differing {r['differ']} outside [lo,hi] {r['outside']}", flush=True)
if nframes > 3:
    print(f"ALL {len(rows)} frame pairs: worst dE*cond {max(r['worst'] for r in rows):.3g} (bound used by the scoring kernel: 1e-16), "
          f"rows with a count outside [lo,hi]: {sum(r['outside'] for r in rows)}, rows whose count differs from the exact one: {sum(r['differ'] for r in rows)}")
T.close(); ctx.close()
