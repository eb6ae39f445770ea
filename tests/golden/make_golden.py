#!/usr/bin/env python3
"""Generate tests/golden/*.npz + e2e text fixtures from the REAL reference build.

Run in the build container only (needs oracle/_ref/, i.e. `make -C oracle ref`, which compiles the
reference sources where they lie under /root/reference):

    python tests/golden/make_golden.py

Every fixture stores the INPUTS next to the reference's OUTPUTS, so the tests that consume them
need neither the reference nor this script.  Inputs are synthetic (the TempleRing dataset is not in
the image) and seeded.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

synth = __import__("importlib").import_module(H.PKG_NAME + ".synth")


def textured_pair(w, h, seed, shift=(0.6, -0.4)):
    """Smooth random texture and a sub-pixel shifted + slightly brightened copy (u8)."""
    rng = np.random.default_rng(seed)
    big = rng.uniform(0, 255, size=(h + 16, w + 16))
    k = np.array([1, 4, 6, 4, 1], float)
    k /= k.sum()
    for _ in range(2):
        big = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, big)
        big = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, big)
    big = (big - big.min()) / (big.max() - big.min()) * 255.0
    ys, xs = np.mgrid[0:h, 0:w].astype(float)

    def samp(dx, dy):
        x = xs + 8 + dx
        y = ys + 8 + dy
        x0 = np.floor(x).astype(int)
        y0 = np.floor(y).astype(int)
        fx, fy = x - x0, y - y0
        return (big[y0, x0] * (1 - fx) * (1 - fy) + big[y0, x0 + 1] * fx * (1 - fy)
                + big[y0 + 1, x0] * (1 - fx) * fy + big[y0 + 1, x0 + 1] * fx * fy)

    a = np.clip(np.rint(samp(0, 0)), 0, 255).astype(np.uint8)
    b = np.clip(np.rint(samp(*shift) * 1.02 + 1.0), 0, 255).astype(np.uint8)
    return a, b


def two_view(n, seed, outlier_frac=0.25, noise=0.3):
    """Pixel correspondences of a random rigid scene seen by two cameras."""
    rng = np.random.default_rng(seed)
    K = synth.K_TEMPLE.copy()
    X = rng.normal(size=(n, 3)) * 0.08 + np.array([0, 0, 0.65])
    R0, t0 = np.eye(3), np.zeros(3)
    R1, t1 = synth.ring_pose(4.0)
    R1 = R1 @ synth.ring_pose(0.0)[0].T
    t1 = np.array([0.05, 0.004, 0.01])

    def proj(R, t):
        Xc = X @ R.T + t
        return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)

    pi = proj(R0, t0) + rng.normal(size=(n, 2)) * noise
    pj = proj(R1, t1) + rng.normal(size=(n, 2)) * noise
    bad = rng.random(n) < outlier_frac
    pj[bad] += rng.uniform(-40, 40, size=(int(bad.sum()), 2))
    return K, pi, pj


def ba_problem(W, P, seed, n_kf=None):
    """n_kf keyframes on the ring, P points each seen in a random subset (>=2) of keyframes."""
    rng = np.random.default_rng(seed)
    n_kf = n_kf or W
    K = synth.K_TEMPLE.copy()
    poses = np.zeros((n_kf, 12))
    Rw, tw = [], []
    for k in range(n_kf):
        R, t = synth.ring_pose(2.0 * k)
        Rw.append(R)
        tw.append(t)
        Rcw = R.T
        C = -Rcw @ t
        # perturb so that BA has something to do
        poses[k, :9] = (Rcw @ H.so3_exp(H.oracle(), "orc", rng.normal(size=3) * 2e-3)).ravel() if k else Rcw.ravel()
        poses[k, 9:] = C + (rng.normal(size=3) * 1e-3 if k else 0)
    X = rng.normal(size=(P, 3)) * 0.06
    obs_ptr, obs_kf, obs_uv = [0], [], []
    for p in range(P):
        m = int(rng.integers(2, n_kf + 1))
        ks = np.sort(rng.choice(n_kf, size=m, replace=False))
        for k in ks:
            Xc = Rw[k] @ X[p] + tw[k]
            uv = np.array([K[0, 0] * Xc[0] / Xc[2] + K[0, 2], K[1, 1] * Xc[1] / Xc[2] + K[1, 2]])
            uv += rng.normal(size=2) * (0.5 if rng.random() > 0.1 else 6.0)  # some Huber-range residuals
            obs_kf.append(k)
            obs_uv.append(uv)
        obs_ptr.append(len(obs_kf))
    return K, poses, X, np.array(obs_ptr, np.int32), np.array(obs_kf, np.int32), np.array(obs_uv)


def main():
    r = H.ref()
    assert r is not None, "oracle/_ref/libsfmref.so missing: run `make -C oracle ref` in the build container"
    if "--e2e-only" not in sys.argv:
        function_vectors(r)
    end_to_end()


def function_vectors(r):
    out = {}

    # ---- images / KLT
    a, b = textured_pair(160, 120, 11)
    out["klt_a"], out["klt_b"] = a, b
    out["ds_a"] = H.downsample2(r, "ref", a)
    odd = a[:119, :157].copy()
    out["ds_odd_in"] = odd
    out["ds_odd"] = H.downsample2(r, "ref", odd)
    corners = H.shi_tomasi(r, "ref", a, 300, 0.01, 6)
    out["shi_corners"] = corners
    out["shi_args"] = np.array([300, 0.01, 6])
    rng = np.random.default_rng(5)
    pts = np.concatenate([corners[:80], rng.uniform(-3, 163, size=(24, 2)),
                          np.array([[0.0, 0.0], [159.0, 119.0], [-50.0, 10.0], [1e6, 5.0], [80.5, 60.25], [5.0, 5.0],
                                    [154.999, 114.5], [2.5, 117.0]])])
    out["klt_pts"] = pts
    for (lv, rad, it) in [(3, 5, 10), (1, 5, 4), (2, 3, 6)]:
        fwd, back, keep = H.klt_track(r, "ref", a, b, lv, rad, it, pts, 1.0)
        out[f"klt_fwd_{lv}_{rad}_{it}"], out[f"klt_back_{lv}_{rad}_{it}"], out[f"klt_keep_{lv}_{rad}_{it}"] = fwd, back, keep
    lk_xy = np.concatenate([corners[:24] + 0.37, np.array([[3.2, 4.9], [0.0, 0.0], [158.7, 60.0], [-9.0, -9.0], [80.0, 118.6]])])
    out["lk_xy"] = lk_xy
    out["lk_step_r5"] = np.array([H.lk_step(r, "ref", a, b, 5, x, y) for x, y in lk_xy])
    out["lk_step_r2"] = np.array([H.lk_step(r, "ref", a, b, 2, x, y) for x, y in lk_xy])

    # ---- stateful tracker over a short synthetic ring sequence (replenish fires)
    seq = synth.make_sequence(5, 160, 120, 0.4, n_blobs=2500, seed=3)
    out["trk_images"] = seq["images"]
    trk_cfg = dict(max_tracks=220, min_tracks=200, quality=0.01, min_distance=5, levels=3, radius=5, iters=10, fb=1.0)
    out["trk_cfg"] = np.array([trk_cfg[k] for k in ("max_tracks", "min_tracks", "quality", "min_distance", "levels", "radius", "iters", "fb")], float)
    T = H.Tracker(r, "ref", **trk_cfg)
    for f in range(5):
        prev, cur, ids = T.step(seq["images"][f])
        txy, tid = T.tracks()
        out[f"trk_prev_{f}"], out[f"trk_cur_{f}"], out[f"trk_ids_{f}"] = prev, cur, ids
        out[f"trk_txy_{f}"], out[f"trk_tid_{f}"] = txy, tid
    T.close()

    # ---- RNG
    for n in (8, 100, 517, 5000, 3):
        out[f"rng_{n}"] = H.uniform_draws(r, "ref", 12345, n, 4096)

    # ---- two-view geometry
    K, pi, pj = two_view(240, 21)
    out["tv_K"], out["tv_pi"], out["tv_pj"] = K, pi, pj
    rc, xi = H.normalize_points(r, "ref", K, pi)
    rc, xj = H.normalize_points(r, "ref", K, pj)
    out["tv_xi"], out["tv_xj"] = xi, xj
    draws = H.uniform_draws(r, "ref", 12345, 240, 8 * 96).reshape(96, 8)
    draws[5] = draws[5][[0, 0, 2, 3, 4, 5, 6, 7]]  # a degenerate octet (duplicate index)
    out["tv_idx8"] = draws
    Es = np.array([H.eight_point(r, "ref", xi, xj, d) for d in draws])
    out["tv_E"] = Es
    out["tv_sampson0"] = np.array([H.sampson(r, "ref", Es[0], xi[i], xj[i]) for i in range(240)])
    for (iters, thr, mi) in [(400, 1e-3, 60), (300, 2e-3, 80), (50, 1e-6, 200)]:
        res = H.find_E_ransac(r, "ref", K, pi, pj, iters, thr, mi)
        tag = f"{iters}_{mi}"
        out[f"rs_ok_{tag}"] = np.array([res["ok"]])
        out[f"rs_R_{tag}"], out[f"rs_t_{tag}"], out[f"rs_inl_{tag}"] = res["R"], res["t"], res["inliers"]
    out["rs_cases"] = np.array([[400, 1e-3, 60], [300, 2e-3, 80], [50, 1e-6, 200]])

    # ---- small eigen / svd / so3 / triangulation
    rng = np.random.default_rng(8)
    for n in (3, 4, 9):
        M = rng.normal(size=(6, n, n))
        M = M + M.transpose(0, 2, 1)
        M[5] = np.diag(np.arange(n, 0, -1.0))
        out[f"jac_in_{n}"] = M
        res = [H.jacobi(r, "ref", m, 120 if n == 9 else 80) for m in M]
        out[f"jac_w_{n}"] = np.array([x[0] for x in res])
        out[f"jac_V_{n}"] = np.array([x[1] for x in res])
    A3 = rng.normal(size=(6, 3, 3))
    A3[4] = np.outer([1, 2, 3], [0.5, -1, 2])  # rank 1
    A3[5] = 0
    out["svd_in"] = A3
    sv = [H.svd3(r, "ref", m) for m in A3]
    out["svd_U"], out["svd_s"], out["svd_V"] = np.array([x[0] for x in sv]), np.array([x[1] for x in sv]), np.array([x[2] for x in sv])
    ws = np.concatenate([rng.normal(size=(6, 3)) * 0.3, np.array([[0, 0, 0], [1e-12, -2e-12, 1e-13], [3.0, 0.1, -0.2]])])
    out["so3_w"] = ws
    Rs = np.array([H.so3_exp(r, "ref", w) for w in ws])
    out["so3_R"] = Rs
    out["so3_log"] = np.array([H.so3_log(r, "ref", R) for R in Rs])
    tri_in, tri_out = [], []
    for k in range(6):
        Rw1, tw1 = synth.ring_pose(0.0)
        Rw2, tw2 = synth.ring_pose(3.0 + k)
        X = rng.normal(size=3) * 0.05
        uv = []
        for (R, t) in ((Rw1, tw1), (Rw2, tw2)):
            Xc = R @ X + t
            uv.append([K[0, 0] * Xc[0] / Xc[2] + K[0, 2] + rng.normal() * 0.3, K[1, 1] * Xc[1] / Xc[2] + K[1, 2] + rng.normal() * 0.3])
        row = np.concatenate([Rw1.T.ravel(), -Rw1.T @ tw1, Rw2.T.ravel(), -Rw2.T @ tw2, uv[0], uv[1]])
        tri_in.append(row)
        tri_out.append(H.triangulate(r, "ref", K, row[0:9], row[9:12], row[12:21], row[21:24], row[24:26], row[26:28]))
    # identical poses -> the reference divides by ~0 (quirk Q7)
    row = tri_in[0].copy()
    row[12:24] = row[0:12]
    tri_in.append(row)
    tri_out.append(H.triangulate(r, "ref", K, row[0:9], row[9:12], row[12:21], row[21:24], row[24:26], row[26:28]))
    out["tri_in"], out["tri_out"] = np.array(tri_in), np.array(tri_out)

    # ---- dense solve
    for n in (6, 36, 60):
        M = rng.normal(size=(n, n))
        A = M @ M.T + np.eye(n) * 1e-3
        A[0, 0] += 1e9
        bb = rng.normal(size=n)
        rc, x = H.solve_gauss(r, "ref", A, bb)
        out[f"sg_A_{n}"], out[f"sg_b_{n}"], out[f"sg_x_{n}"], out[f"sg_rc_{n}"] = A, bb, x, np.array([rc])
    A = rng.normal(size=(5, 5))
    A[3] = A[1] * 2.0  # exactly singular -> reference throws
    rc, x = H.solve_gauss(r, "ref", A, np.ones(5))
    out["sg_A_sing"], out["sg_rc_sing"] = A, np.array([rc])
    A = rng.normal(size=(7, 7))  # general non-symmetric with pivoting
    bb = rng.normal(size=7)
    rc, x = H.solve_gauss(r, "ref", A, bb)
    out["sg_A_7"], out["sg_b_7"], out["sg_x_7"] = A, bb, x

    # ---- bundle adjustment (finite inputs; the map iteration order is part of the answer)
    for (W, P, nk) in [(2, 30, 2), (6, 80, 8), (10, 120, 12)]:
        K, poses, X, optr, okf, ouv = ba_problem(W, P, 100 + W, nk)
        tag = f"{W}_{P}"
        out[f"ba_K_{tag}"], out[f"ba_poses_{tag}"], out[f"ba_X_{tag}"] = K, poses, X
        out[f"ba_optr_{tag}"], out[f"ba_okf_{tag}"], out[f"ba_ouv_{tag}"] = optr, okf, ouv
        for iters in (1, 5):
            out[f"ba_out_{tag}_{iters}"] = H.bundle_adjust_window(r, "ref", K, poses, X, optr, okf, ouv, W, iters, 600, 3.0, 1e-3)
        out[f"ba_out_{tag}_cap"] = H.bundle_adjust_window(r, "ref", K, poses, X, optr, okf, ouv, W, 2, P // 2, 3.0, 1e-3)
    out["ba_cases"] = np.array([[2, 30, 2], [6, 80, 8], [10, 120, 12]])
    for n in (5, 13, 100, 700):
        out[f"maporder_{n}"] = H.map_iteration_order(r, "ref", n)

    # ---- pose graph + descriptor
    nk = 9
    Rs = np.array([synth.ring_pose(5.0 * k)[0].T for k in range(nk)])
    Cs = np.array([-synth.ring_pose(5.0 * k)[0].T @ synth.ring_pose(5.0 * k)[1] for k in range(nk)]) + rng.normal(size=(nk, 3)) * 0.01
    ei = np.array(list(range(nk - 1)) + [0], np.int32)
    ej = np.array(list(range(1, nk)) + [nk - 1], np.int32)
    eR = np.array([H.so3_exp(r, "ref", rng.normal(size=3) * 0.05) for _ in ei])
    et = rng.normal(size=(len(ei), 3))
    et /= np.linalg.norm(et, axis=1, keepdims=True)
    lp = np.array([0] * (nk - 1) + [1], np.int32)
    ok, c2 = H.posegraph(r, "ref", Rs, Cs, ei, ej, eR, et, lp)
    out["pg_R"], out["pg_C"], out["pg_ei"], out["pg_ej"], out["pg_eR"], out["pg_et"], out["pg_loop"] = Rs, Cs, ei, ej, eR, et, lp
    out["pg_ok"], out["pg_out"] = np.array([ok]), c2
    out["desc_a"] = H.global_desc(r, "ref", a)
    out["desc_odd"] = H.global_desc(r, "ref", odd)

    np.savez_compressed(os.path.join(HERE, "hotpath.npz"), **out)
    print("wrote hotpath.npz with", len(out), "arrays")



def end_to_end():
    """the reference CLI on small synthetic datasets (e2e_loop revisits earlier views so that loop closure,
    the pose-graph solve and the second BA pass fire: T:1822-1866)"""
    cli = H.ref_cli()
    only = [a for a in sys.argv[1:] if a.startswith("e2e_")]
    for name, (frames, deg, cfgover, size, nb, seed, angles) in {
        "e2e_small": (8, 0.4, {"klt": {"max_tracks": 400, "min_tracks": 250, "min_distance": 5}}, (160, 120), 2500, 9, None),
        "e2e_keyframes": (10, 0.25, {"klt": {"max_tracks": 500, "min_tracks": 300, "min_distance": 4},
                                     "keyframe": {"min_inliers": 120, "parallax_px": 3.0, "min_gap": 2},
                                     "ba": {"window": 4, "iters": 3, "max_points": 150}}, (160, 120), 2500, 9, None),
        "e2e_loop": (14, 0.3, {"klt": {"max_tracks": 900, "min_tracks": 600, "min_distance": 5},
                               "keyframe": {"min_inliers": 100, "parallax_px": 1.0, "min_gap": 1},
                               "ba": {"window": 4, "iters": 3, "max_points": 200}}, (320, 240), 6000, 13,
                     [0, 0.3, 0.6, 0.9, 1.2, 1.5, 1.8, 1.5, 1.2, 0.9, 0.6, 0.3, 0.0, 0.3]),
    }.items():
        if only and name not in only:
            continue
        seq = synth.make_sequence(frames, size[0], size[1], deg, n_blobs=nb, seed=seed, angles=angles)
        cfg = {"common": {"system": {"frames": frames}, "klt": cfgover.get("klt", {}), "keyframe": cfgover.get("keyframe", {})},
               "cpp": {"ba": cfgover.get("ba", {})}}
        with tempfile.TemporaryDirectory() as td:
            synth.write_dataset(td, seq)
            with open(os.path.join(td, "cfg.json"), "w") as f:
                json.dump(cfg, f)
            p = subprocess.run([cli, td, os.path.join(td, "out"), "--config", os.path.join(td, "cfg.json")],
                               capture_output=True, text=True, cwd=td)
            assert p.returncode == 0, p.stderr
            stdout = p.stdout.replace(os.path.join(td, "out"), "<OUT>")
            files = {fn: open(os.path.join(td, "out", fn)).read() for fn in
                     ("keyframes_camera_centers.csv", "posegraph_edges.csv", "templeRing_sparse_points.ply")}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), images=seq["images"], K=seq["K"], R=seq["R"], t=seq["t"],
                            lat=seq["lat"], lon=seq["lon"], names=np.array(seq["names"]), config=np.array(json.dumps(cfg)),
                            stdout=np.array(stdout), **{k.replace(".", "_"): np.array(v) for k, v in files.items()})
        print(name, "->", stdout.strip().splitlines()[-4:])


if __name__ == "__main__":
    main()
