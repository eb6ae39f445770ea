#!/usr/bin/env python3
"""Generate tests/golden/ransac_degenerate.npz from the REAL reference build (oracle/_ref).

find_E_ransac samples its octets WITH replacement (T:665), so some octets repeat an index; the
design matrix then has a null space of dimension >= 2 and the hypothesis the reference scores is
whatever its libm Jacobi lands on.  This script searches seeded synthetic two-view problems with few
correspondences (so that many octets are degenerate) for the cases a ranking by approximate
hypotheses would get wrong:

  deg_wins   the reference's winning iteration is a repeated-index octet
  deg_ties   a repeated-index iteration reaches the maximal count EARLIER than a clean iteration
             with the same count (strict '>' at T:673 keeps the earlier one)
  clean_wins a clean iteration wins and a repeated-index one ties it LATER

Stored per case: inputs (K, pi, pj, iters, thr, min_inliers), the reference's per-iteration
hypotheses and inlier counts (ref_eight_point_E / ref_sampson_err on the library's own sample
stream), and the full ref_find_E_ransac output.  Run in the build container only:

    python tests/golden/make_ransac_degenerate_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

synth = __import__("importlib").import_module(H.PKG_NAME + ".synth")


def scene(n, seed, outlier_frac, noise):
    rng = np.random.default_rng(seed)
    K = synth.K_TEMPLE.copy()
    X = rng.normal(size=(n, 3)) * 0.08 + np.array([0, 0, 0.65])
    R1, _ = synth.ring_pose(5.0)
    R1 = R1 @ synth.ring_pose(0.0)[0].T
    t1 = np.array([0.06, -0.003, 0.012])

    def proj(R, t):
        Xc = X @ R.T + t
        return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)

    pi = proj(np.eye(3), np.zeros(3)) + rng.normal(size=(n, 2)) * noise
    pj = proj(R1, t1) + rng.normal(size=(n, 2)) * noise
    bad = rng.random(n) < outlier_frac
    pj[bad] += rng.uniform(-60, 60, size=(int(bad.sum()), 2))
    return K, pi, pj


def reference_iterations(r, K, pi, pj, iters, thr):
    n = len(pi)
    _, xi = H.normalize_points(r, "ref", K, pi)
    _, xj = H.normalize_points(r, "ref", K, pj)
    idx8 = H.uniform_draws(r, "ref", 12345, n, 8 * iters).reshape(iters, 8)
    Es = np.array([H.eight_point(r, "ref", xi, xj, d) for d in idx8])
    counts = np.array([sum(H.sampson(r, "ref", E, xi[i], xj[i]) < thr for i in range(n)) for E in Es], np.int32)
    deg = np.array([len(set(d)) < 8 for d in idx8])
    return xi, xj, idx8, Es, counts, deg


def main():
    r = H.ref()
    assert r is not None, "oracle/_ref/libsfmref.so missing: run `make -C oracle ref` in the build container"
    want = {"deg_wins": None, "deg_ties": None, "clean_wins": None}
    iters, thr, min_inl = 400, 5e-6, 8
    for seed in range(2000):
        if all(v is not None for v in want.values()):
            break
        n = 18 + seed % 30
        K, pi, pj = scene(n, 1000 + seed, outlier_frac=0.35, noise=0.8)
        xi, xj, idx8, Es, counts, deg = reference_iterations(r, K, pi, pj, iters, thr)
        best = int(np.argmax(counts))  # first maximum == lowest iteration
        ties = np.nonzero(counts == counts[best])[0]
        kind = None
        if deg[best] and len(ties) == 1:
            kind = "deg_wins"
        elif deg[best] and (~deg[ties[1:]]).any():
            kind = "deg_ties"
        elif not deg[best] and deg[ties[1:]].any():
            kind = "clean_wins"
        if kind is None or want[kind] is not None or counts[best] < min_inl:
            continue
        res = H.find_E_ransac(r, "ref", K, pi, pj, iters, thr, min_inl)
        assert res["ok"] == 1 and len(res["inliers"]) == counts[best]
        want[kind] = dict(K=K, pi=pi, pj=pj, xi=xi, xj=xj, idx8=idx8.astype(np.int32), E=Es, counts=counts, deg=deg,
                          args=np.array([iters, thr, min_inl]), best=np.array([best]), R=res["R"], t=res["t"], inl=res["inliers"])
        print(kind, "seed", seed, "n", n, "best iter", best, "count", counts[best], "ties", ties[:6], "degenerate share", deg.mean())
    assert all(v is not None for v in want.values()), {k: v is not None for k, v in want.items()}
    out = {f"{kind}_{k}": v for kind, d in want.items() for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, "ransac_degenerate.npz"), **out)
    print("wrote ransac_degenerate.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
