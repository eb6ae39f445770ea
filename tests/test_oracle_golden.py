"""CPU suite: the oracle restatement against the committed golden vectors (generated from the real
reference build by tests/golden/make_golden.py).  Bit-exact everywhere (FP64 compared as uint64)."""
import json
import os

import numpy as np
import pytest

import helpers as H

O = H.oracle()


def test_downsample(golden):
    assert np.array_equal(H.downsample2(O, "orc", golden["klt_a"]), golden["ds_a"])
    assert np.array_equal(H.downsample2(O, "orc", golden["ds_odd_in"]), golden["ds_odd"])


def test_shi_tomasi(golden):
    mc, q, md = golden["shi_args"]
    got = H.shi_tomasi(O, "orc", golden["klt_a"], int(mc), float(q), int(md))
    H.assert_bits_equal(got, golden["shi_corners"], "shi corners")


@pytest.mark.parametrize("radius", [5, 2])
def test_lk_step(golden, radius):
    got = np.array([H.lk_step(O, "orc", golden["klt_a"], golden["klt_b"], radius, x, y) for x, y in golden["lk_xy"]])
    H.assert_bits_equal(got, golden[f"lk_step_r{radius}"], "lk_step")


@pytest.mark.parametrize("cfg", [(3, 5, 10), (1, 5, 4), (2, 3, 6)])
def test_klt_track(golden, cfg):
    lv, rad, it = cfg
    fwd, back, keep = H.klt_track(O, "orc", golden["klt_a"], golden["klt_b"], lv, rad, it, golden["klt_pts"], 1.0)
    tag = f"{lv}_{rad}_{it}"
    H.assert_bits_equal(fwd, golden[f"klt_fwd_{tag}"], "fwd")
    H.assert_bits_equal(back, golden[f"klt_back_{tag}"], "back")
    assert np.array_equal(keep, golden[f"klt_keep_{tag}"])


def test_tracker_sequence(golden):
    c = golden["trk_cfg"]
    T = H.Tracker(O, "orc", int(c[0]), int(c[1]), float(c[2]), int(c[3]), int(c[4]), int(c[5]), int(c[6]), float(c[7]))
    for f in range(5):
        prev, cur, ids = T.step(golden["trk_images"][f])
        txy, tid = T.tracks()
        assert np.array_equal(ids, golden[f"trk_ids_{f}"])
        H.assert_bits_equal(prev, golden[f"trk_prev_{f}"], f"prev {f}")
        H.assert_bits_equal(cur, golden[f"trk_cur_{f}"], f"cur {f}")
        assert np.array_equal(tid, golden[f"trk_tid_{f}"])
        H.assert_bits_equal(txy, golden[f"trk_txy_{f}"], f"tracks {f}")
    # the sequence must have exercised the replenish branch (T:374-389)
    assert len(golden["trk_tid_4"]) > len(golden["trk_ids_4"])


@pytest.mark.parametrize("n", [8, 100, 517, 5000, 3])
def test_rng(golden, n):
    assert np.array_equal(H.uniform_draws(O, "orc", 12345, n, 4096), golden[f"rng_{n}"])


def test_normalize_and_eight_point(golden):
    rc, xi = H.normalize_points(O, "orc", golden["tv_K"], golden["tv_pi"])
    rc2, xj = H.normalize_points(O, "orc", golden["tv_K"], golden["tv_pj"])
    assert rc == 0 and rc2 == 0
    H.assert_bits_equal(xi, golden["tv_xi"], "xi")
    H.assert_bits_equal(xj, golden["tv_xj"], "xj")
    Es = np.array([H.eight_point(O, "orc", xi, xj, d) for d in golden["tv_idx8"]])
    H.assert_bits_equal(Es, golden["tv_E"], "E")
    s0 = np.array([H.sampson(O, "orc", Es[0], xi[i], xj[i]) for i in range(len(xi))])
    H.assert_bits_equal(s0, golden["tv_sampson0"], "sampson")
    assert H.normalize_points(O, "orc", np.zeros((3, 3)), golden["tv_pi"])[0] == 1  # "Singular K" (T:474)


def test_find_E_ransac(golden):
    for iters, thr, mi in golden["rs_cases"]:
        tag = f"{int(iters)}_{int(mi)}"
        r = H.find_E_ransac(O, "orc", golden["tv_K"], golden["tv_pi"], golden["tv_pj"], int(iters), float(thr), int(mi))
        assert r["ok"] == int(golden[f"rs_ok_{tag}"][0])
        if r["ok"]:
            assert np.array_equal(r["inliers"], golden[f"rs_inl_{tag}"])
            H.assert_bits_equal(r["R"], golden[f"rs_R_{tag}"], "R")
            H.assert_bits_equal(r["t"], golden[f"rs_t_{tag}"], "t")
    # fewer than 8 points -> nullopt (T:648)
    assert H.find_E_ransac(O, "orc", golden["tv_K"], golden["tv_pi"][:7], golden["tv_pj"][:7], 10, 1e-3, 1)["ok"] == 0


@pytest.mark.parametrize("kind", ["deg_wins", "deg_ties", "clean_wins"])
def test_find_E_ransac_degenerate_octets(kind):
    """tests/golden/ransac_degenerate.npz (reference output): repeated-index octets win / tie earlier / tie later."""
    g = np.load(os.path.join(H.GOLDEN, "ransac_degenerate.npz"))
    iters, thr, mi = g[f"{kind}_args"]
    xi, xj, idx8 = g[f"{kind}_xi"], g[f"{kind}_xj"], g[f"{kind}_idx8"]
    assert np.array_equal(H.uniform_draws(O, "orc", 12345, len(xi), 8 * int(iters)).reshape(-1, 8), idx8)
    Es = np.array([H.eight_point(O, "orc", xi, xj, d) for d in idx8])
    H.assert_bits_equal(Es, g[f"{kind}_E"], "E of every iteration (degenerate ones included)")
    cnt = np.zeros(len(idx8), np.int32)
    O.call("orc_ransac_counts", None, H.f64(xi), H.f64(xj), len(xi), H.f64(Es), len(idx8), float(thr), cnt)
    assert np.array_equal(cnt, g[f"{kind}_counts"])
    r = H.find_E_ransac(O, "orc", g[f"{kind}_K"], g[f"{kind}_pi"], g[f"{kind}_pj"], int(iters), float(thr), int(mi))
    assert r["ok"] == 1 and r["best_iter"] == int(g[f"{kind}_best"][0])
    assert np.array_equal(r["inliers"], g[f"{kind}_inl"])
    H.assert_bits_equal(r["R"], g[f"{kind}_R"], "R")
    H.assert_bits_equal(r["t"], g[f"{kind}_t"], "t")


@pytest.mark.parametrize("n", [3, 4, 9])
def test_jacobi(golden, n):
    for k, m in enumerate(golden[f"jac_in_{n}"]):
        w, V = H.jacobi(O, "orc", m, 120 if n == 9 else 80)
        H.assert_bits_equal(w, golden[f"jac_w_{n}"][k], "w")
        H.assert_bits_equal(V, golden[f"jac_V_{n}"][k], "V")


def test_svd3_so3_triangulate(golden):
    for k, m in enumerate(golden["svd_in"]):
        U, s, V = H.svd3(O, "orc", m)
        H.assert_bits_equal(U, golden["svd_U"][k], "U")
        H.assert_bits_equal(s, golden["svd_s"][k], "s")
        H.assert_bits_equal(V, golden["svd_V"][k], "V")
    for k, w in enumerate(golden["so3_w"]):
        R = H.so3_exp(O, "orc", w)
        H.assert_bits_equal(R, golden["so3_R"][k], "exp")
        H.assert_bits_equal(H.so3_log(O, "orc", R), golden["so3_log"][k], "log")
    K = golden["tv_K"]
    for k, row in enumerate(golden["tri_in"]):
        X = H.triangulate(O, "orc", K, row[0:9], row[9:12], row[12:21], row[21:24], row[24:26], row[26:28])
        H.assert_bits_equal(X, golden["tri_out"][k], f"tri {k}")


def test_solve_gauss(golden):
    for n in (6, 36, 60, 7):
        rc, x = H.solve_gauss(O, "orc", golden[f"sg_A_{n}"], golden[f"sg_b_{n}"])
        assert rc == 0
        H.assert_bits_equal(x, golden[f"sg_x_{n}"], f"x{n}")
    rc, _ = H.solve_gauss(O, "orc", golden["sg_A_sing"], np.ones(5))
    assert rc == int(golden["sg_rc_sing"][0]) == 1


def test_bundle_adjust(golden):
    for W, P, nk in golden["ba_cases"]:
        tag = f"{int(W)}_{int(P)}"
        args = (golden[f"ba_K_{tag}"], golden[f"ba_poses_{tag}"], golden[f"ba_X_{tag}"], golden[f"ba_optr_{tag}"],
                golden[f"ba_okf_{tag}"], golden[f"ba_ouv_{tag}"])
        for iters in (1, 5):
            got = H.bundle_adjust_window(O, "orc", *args, int(W), iters, 600, 3.0, 1e-3)
            H.assert_bits_equal(got, golden[f"ba_out_{tag}_{iters}"], f"BA {tag} it{iters}")
        got = H.bundle_adjust_window(O, "orc", *args, int(W), 2, int(P) // 2, 3.0, 1e-3)
        H.assert_bits_equal(got, golden[f"ba_out_{tag}_cap"], f"BA {tag} capped")
    for n in (5, 13, 100, 700):
        assert np.array_equal(H.map_iteration_order(O, "orc", n), golden[f"maporder_{n}"])


def test_posegraph_and_descriptor(golden):
    ok, c = H.posegraph(O, "orc", golden["pg_R"], golden["pg_C"], golden["pg_ei"], golden["pg_ej"], golden["pg_eR"],
                        golden["pg_et"], golden["pg_loop"])
    assert ok == int(golden["pg_ok"][0])
    H.assert_bits_equal(c, golden["pg_out"], "pose graph")
    assert np.array_equal(H.global_desc(O, "orc", golden["klt_a"]).view(np.uint32), golden["desc_a"].view(np.uint32))
    assert np.array_equal(H.global_desc(O, "orc", golden["ds_odd_in"]).view(np.uint32), golden["desc_odd"].view(np.uint32))


def check_e2e_against_reference(g, log, out):
    """Compare a pipeline run with the reference CLI's golden output.

    Everything the reference computes with defined behaviour must be byte-identical: the whole stdout
    (per-frame keyframe / map-point counts), posegraph_edges.csv (RANSAC R,t,inliers per keyframe pair)
    and the id / frame / image / lat / lon columns of keyframes_camera_centers.csv plus the PLY header.
    The x,y,z columns and PLY coordinates are NOT compared with the reference: T:1809 reads
    kfs[idl] one past the end of the vector for the keyframe under construction (it is pushed at
    T:1815), so the reference triangulates against heap garbage and its map points / BA-refined
    centres differ from run environment to run environment (quirk Q12 in DESIGN.md; the goldens show
    inf/nan in one scenario, 1e8-sized junk in another, zeros in a third).
    """
    assert log.replace(out, "<OUT>") == str(g["stdout"])
    assert open(os.path.join(out, "posegraph_edges.csv")).read() == str(g["posegraph_edges_csv"])
    got = open(os.path.join(out, "keyframes_camera_centers.csv")).read().splitlines()
    exp = str(g["keyframes_camera_centers_csv"]).splitlines()
    assert len(got) == len(exp) and got[0] == exp[0] and got[1] == exp[1]  # header + gauge keyframe (0,0,0)
    for a, b in zip(got[2:], exp[2:]):
        fa, fb = a.split(","), b.split(",")
        assert fa[:3] + fa[6:] == fb[:3] + fb[6:]
    gp = open(os.path.join(out, "templeRing_sparse_points.ply")).read().splitlines()
    ep = str(g["templeRing_sparse_points_ply"]).splitlines()
    assert gp[:7] == ep[:7] and len(gp) == len(ep)


@pytest.mark.parametrize("name", ["e2e_small", "e2e_keyframes", "e2e_loop"])
def test_pipeline_end_to_end(name, tmp_path):
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    cfg = H.pipe_cfg_from_json(json.loads(str(g["config"])))
    out = str(tmp_path / "out")
    rc, log, nk, npnt = H.orc_pipeline_run(g["images"], [str(s) for s in g["names"]], g["K"], g["lat"], g["lon"], cfg, out)
    assert rc == 0
    check_e2e_against_reference(g, log, out)
    # defined-behaviour part must at least be finite here
    rows = open(os.path.join(out, "keyframes_camera_centers.csv")).read().splitlines()[1:]
    assert all(np.isfinite([float(v) for v in r.split(",")[3:6]]).all() for r in rows)
    if name == "e2e_loop":  # the scenario must exercise loop closure + pose graph (is_loop == 1 rows)
        assert sum(l.endswith(",1") for l in str(g["posegraph_edges_csv"]).splitlines()) >= 3
