"""GPU suite: every HIP kernel through the C ABI against the oracle on the same inputs, and against
the committed golden vectors (which came from the real reference build).  Bit-exact."""
import importlib
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
O = H.oracle()
capi = importlib.import_module(H.PKG_NAME + ".capi")
synth = importlib.import_module(H.PKG_NAME + ".synth")


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def test_device_arithmetic_matches_host(ctx):
    """IEEE division / sqrt and the glibc-compatible hypot on gfx950 vs the host (bitwise)."""
    rng = np.random.default_rng(0)
    x = np.ldexp(rng.uniform(-1, 1, 400000), rng.integers(-30, 30, 400000))
    y = np.ldexp(rng.uniform(-1, 1, 400000), rng.integers(-30, 30, 400000))
    y[::17] = x[::17] * rng.uniform(-1, 1, x[::17].size)
    x[5], y[5] = 0.0, 0.0
    x[6], y[6] = np.inf, 1.0
    x[7], y[7] = np.nan, 1.0
    H.assert_bits_equal(ctx.debug_hypot(x, y), np.hypot(x, y), "hypot")
    d, s = ctx.debug_divsqrt(x, y)
    with np.errstate(all="ignore"):
        H.assert_bits_equal(d, x / y, "div")
        H.assert_bits_equal(s, np.sqrt(np.abs(x)), "sqrt")


@pytest.mark.parametrize("shape", [(120, 160), (119, 157), (480, 640), (5, 7)])
def test_pyramid(ctx, shape, golden):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape, dtype=np.uint8) if shape != (120, 160) else golden["klt_a"]
    pyr = ctx.pyramid(img, 3)
    cur = img
    assert np.array_equal(pyr.level(0), img)
    for l in (1, 2):
        cur = H.downsample2(O, "orc", cur) if min(cur.shape) >= 2 else cur[:0, :0]
        got = pyr.level(l)
        assert got.shape == cur.shape and np.array_equal(got, cur), f"level {l}"
    if shape == (120, 160):
        assert np.array_equal(pyr.level(1), golden["ds_a"])


@pytest.mark.parametrize("shape", [(120, 160), (97, 131), (480, 640)])
def test_shi_score_and_candidates(ctx, shape, golden):
    if shape == (120, 160):
        img = golden["klt_a"]
    else:
        img = synth.make_sequence(1, shape[1], shape[0], 0.3, n_blobs=3000, seed=shape[0])["images"][0]
    pyr = ctx.pyramid(img, 1)
    score, mx = ctx.shi_score(pyr)
    exp = np.zeros(shape)
    O.call("orc_shi_score", None, H.u8(img), shape[1], shape[0], exp)
    H.assert_bits_equal(score, exp, "score map")
    assert mx == exp.max()
    xs, ys, sc, n, mx2 = ctx.shi_candidates(pyr, 0.01)
    thr = exp.max() * 0.01
    yy, xx = np.nonzero(exp >= thr)  # row-major order
    assert n == len(xx) and mx2 == mx
    assert np.array_equal(xs, xx) and np.array_equal(ys, yy)
    H.assert_bits_equal(sc, exp[yy, xx], "candidate scores")


@pytest.mark.parametrize("cfg", [(3, 5, 10), (1, 5, 4), (2, 3, 6)])
def test_klt_golden(ctx, golden, cfg):
    lv, rad, it = cfg
    pa, pb = ctx.pyramid(golden["klt_a"], lv), ctx.pyramid(golden["klt_b"], lv)
    fwd, back, keep, steps = ctx.klt_track(pa, pb, golden["klt_pts"], lv, rad, it, 1.0)
    tag = f"{lv}_{rad}_{it}"
    H.assert_bits_equal(fwd, golden[f"klt_fwd_{tag}"], "fwd")
    H.assert_bits_equal(back, golden[f"klt_back_{tag}"], "back")
    assert np.array_equal(keep, golden[f"klt_keep_{tag}"])
    assert 0 < steps <= 2 * lv * it * len(golden["klt_pts"])


@pytest.mark.parametrize("radius", [5, 2])
def test_lk_single_step_golden(ctx, golden, radius):
    """levels=1, iters=1: forward result - input == lk_step (T:424-460)."""
    pa, pb = ctx.pyramid(golden["klt_a"], 1), ctx.pyramid(golden["klt_b"], 1)
    xy = golden["lk_xy"]
    fwd, _, _, _ = ctx.klt_track(pa, pb, xy, 1, radius, 1, 1.0)
    exp = (xy + 0.0)  # p = (pl + dl) * 1 with dl = step
    exp = np.stack([(xy[:, 0] + golden[f"lk_step_r{radius}"][:, 0]) * 1.0, (xy[:, 1] + golden[f"lk_step_r{radius}"][:, 1]) * 1.0], 1)
    H.assert_bits_equal(fwd, exp, "single lk_step")


def test_klt_synthetic_vs_oracle(ctx):
    seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=7)
    a, b = seq["images"]
    pts = H.shi_tomasi(O, "orc", a, 400, 0.01, 8)
    rng = np.random.default_rng(3)
    pts = np.concatenate([pts, rng.uniform(-20, 660, size=(40, 2)), pts[:20] + rng.uniform(-0.5, 0.5, (20, 2))])
    pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
    fwd, back, keep, steps = ctx.klt_track(pa, pb, pts)
    efwd, eback, ekeep = H.klt_track(O, "orc", a, b, 3, 5, 10, pts, 1.0)
    H.assert_bits_equal(fwd, efwd, "fwd")
    H.assert_bits_equal(back, eback, "back")
    assert np.array_equal(keep, ekeep)
    assert keep.sum() > 100  # the case must actually track something


def test_tracker_step_vs_reference_goldens(ctx, golden):
    """KLTTracker::{reset,step} (T:322-391) through the host seam against the state the compiled reference produced
    (trk_* in hotpath.npz): StepOut ids / prev / cur of every frame and the live tracks after replenishment."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    c = golden["trk_cfg"]
    imgs = golden["trk_images"]
    T = pipe.Tracker(ctx, imgs.shape[2], imgs.shape[1], int(c[0]), int(c[1]), float(c[2]), int(c[3]), int(c[4]), int(c[5]), int(c[6]), float(c[7]))
    for f in range(5):
        prev, cur, ids = T.step(imgs[f])
        txy, tid = T.tracks()
        assert np.array_equal(ids, golden[f"trk_ids_{f}"]), f
        H.assert_bits_equal(prev, golden[f"trk_prev_{f}"], f"prev {f}")
        H.assert_bits_equal(cur, golden[f"trk_cur_{f}"], f"cur {f}")
        assert np.array_equal(tid, golden[f"trk_tid_{f}"]), f
        H.assert_bits_equal(txy, golden[f"trk_txy_{f}"], f"tracks {f}")
    T.close()


def test_c5_image_size_1920x1080_vs_oracle(ctx):
    """BASELINE config 5's image size: pyramid, Shi-Tomasi score map, and the tracker seam (detector pick + KLT fwd/bwd +
    replenish) on 1920x1080 frames with 5000 tracks, against the oracle."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    seq = synth.make_sequence(3, 1920, 1080, 0.003, n_blobs=120000, seed=3, shell_scale=2.2)  # the reference LK overshoots: sub-pixel flow only
    a = seq["images"][0]
    pyr = ctx.pyramid(a, 4)
    cur = a
    for l in (1, 2, 3):
        cur = H.downsample2(O, "orc", cur)
        assert np.array_equal(pyr.level(l), cur), f"level {l}"
    score, mx = ctx.shi_score(pyr)
    exp = np.zeros(a.shape)
    O.call("orc_shi_score", None, H.u8(a), a.shape[1], a.shape[0], exp)
    H.assert_bits_equal(score, exp, "1080p score map")
    assert mx == exp.max()
    kw = dict(max_tracks=5000, min_tracks=4800, quality=0.01, min_distance=8, levels=3, radius=5, iters=10, fb=1.0)
    Tg = pipe.Tracker(ctx, 1920, 1080, **kw)
    To = H.Tracker(O, "orc", **kw)
    for f in range(3):
        gp, gc, gi = Tg.step(seq["images"][f])
        op, oc, oi = To.step(seq["images"][f])
        assert np.array_equal(gi, oi), f
        H.assert_bits_equal(gp, op, f"prev {f}")
        H.assert_bits_equal(gc, oc, f"cur {f}")
        gxy, gid = Tg.tracks()
        oxy, oid = To.tracks()
        assert np.array_equal(gid, oid), f
        H.assert_bits_equal(gxy, oxy, f"tracks {f}")
        assert len(gid) >= 4000, len(gid)
    assert len(gi) >= 3000  # the frames must really be tracked, not only re-seeded
    Tg.close()
    To.close()


@pytest.mark.parametrize("min_distance", [1, 2, 3, 5, 8, 13, 16])
def test_corner_pick_for_every_min_distance_vs_oracle(ctx, min_distance):
    """shi_tomasi's greedy pick (T:286-300) through the tracker seam for the disc radii the device fixpoint supports:
    min_distance 1 takes the plain-scan path of k_shi_round (no neighbours at all), 2..16 the accepted-list / direct-neighbour
    path with discs of 1..15 pixels.  Reset (frame 0) and one replenish (frame 1), tracks bit-equal to the oracle's."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    seq = synth.make_sequence(2, 320, 240, 0.4, n_blobs=5000, seed=40 + min_distance)
    kw = dict(max_tracks=4000, min_tracks=3900, quality=0.01, min_distance=min_distance, levels=3, radius=5, iters=10, fb=1.0)
    Tg = pipe.Tracker(ctx, 320, 240, **kw)
    To = H.Tracker(O, "orc", **kw)
    for f in range(2):
        gp, gc, gi = Tg.step(seq["images"][f])
        op, oc, oi = To.step(seq["images"][f])
        assert np.array_equal(gi, oi), (min_distance, f)
        H.assert_bits_equal(gc, oc, f"cur {f}")
        gxy, gid = Tg.tracks()
        oxy, oid = To.tracks()
        assert np.array_equal(gid, oid), (min_distance, f)
        H.assert_bits_equal(gxy, oxy, f"tracks min_distance={min_distance} frame {f}")
    assert len(gid) > 0
    Tg.close()
    To.close()


@pytest.mark.parametrize("mode", ["tile", "tile,2", "tile,5", "sweeps"])
@pytest.mark.parametrize("min_distance", [2, 8, 16])
def test_corner_schedules_give_the_oracles_pick(ctx, mode, min_distance, monkeypatch):
    """Every schedule of the corner fixpoint (tile-resident kernel with 2 / 3 / 5 passes, sweep schedule) leaves a different set of
    undecided candidates to the host resolver and must end in the same pick: tracks after a reset and a replenish, bit-equal to
    the oracle's (T:286-300), at 640x480 where the tile schedule is not the default."""
    monkeypatch.setenv("SFMX_SHI_MODE", mode)
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    seq = synth.make_sequence(2, 640, 480, 0.4, n_blobs=15000, seed=70 + min_distance)
    kw = dict(max_tracks=3000, min_tracks=2900, quality=0.01, min_distance=min_distance, levels=3, radius=5, iters=10, fb=1.0)
    Tg = pipe.Tracker(ctx, 640, 480, **kw)
    To = H.Tracker(O, "orc", **kw)
    for f in range(2):
        Tg.step(seq["images"][f])
        To.step(seq["images"][f])
        gxy, gid = Tg.tracks()
        oxy, oid = To.tracks()
        assert np.array_equal(gid, oid), (mode, min_distance, f)
        H.assert_bits_equal(gxy, oxy, f"tracks mode={mode} min_distance={min_distance} frame {f}")
    assert len(gid) > 0
    Tg.close()
    To.close()


def test_fp64_matrix_core_adds_in_order():
    """What the KLT kernel's ordered sums rest on (csrc/hip/klt.hip, MSUM): v_mfma_f64_4x4x4 applies its four k-terms as fused
    multiply-adds in ascending k, each rounded to FP64 -- with B = 1.0 four of the reference's additions in order -- keeps
    denormal addends and never produces -0 from +0.  Measured by the stand-alone probe (tools/probes/mfma_f64_order.hip)."""
    import subprocess
    exe = os.path.join(H.ROOT, H.PKG_NAME, "_build", "probes", "mfma_f64_order")
    if not os.path.exists(exe):  # normally built by __graft_entry__.build() / make all
        subprocess.run(["make", "-C", os.path.join(H.ROOT, H.PKG_NAME, "csrc"), "probes"], check=True, capture_output=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    import re
    m = re.search(r"4x4x4f64 rounding over (\d+) sums.*sequential-ascending (\d+), sequential-descending (\d+), wide-then-round (\d+), other (\d+)", p.stdout)
    assert m, p.stdout[-2000:]
    n, seq, rev, wide, other = map(int, m.groups())
    assert n >= 1000000 and seq == n and rev == wide == other == 0, m.group(0)
    m16 = re.search(r"16x16x4f64 rounding: sequential k=0..3 (\d+), sequential k=3..0 (\d+), wide-then-round (\d+), other (\d+)", p.stdout)
    assert m16 and int(m16.group(1)) > 0 and m16.group(2) == m16.group(3) == m16.group(4) == "0", p.stdout[-2000:]
    assert "denormal addend kept: yes" in p.stdout and "0 of 64 results have the sign bit" in p.stdout, p.stdout[-2000:]
    # operand / result lanes the kernel assumes: A[b][i][k] in lane 16 k + 4 b + i, D[b][i][j] in lane 16 i + 4 b + j
    assert re.search(r"^A 1: B0->D16, B1->D17, B2->D18, B3->D19,$", p.stdout, re.M) and re.search(r"^A16: B16->D0, B17->D1, B18->D2, B19->D3,$", p.stdout, re.M)
    assert re.search(r"^A 4: B4->D4, B5->D5, B6->D6, B7->D7,$", p.stdout, re.M)


@pytest.mark.parametrize("radius", [1, 2, 4, 5, 7])
def test_klt_sum_schedules_vs_oracle(ctx, radius, monkeypatch):
    """Every schedule of the ordered sums -- FP64 matrix core (default from radius 2), VALU chains, each with and without the
    pipelined first half (radius 4 / 5) -- against the oracle, bit for bit: interior tracks, tracks on and beyond the image
    border, far-away and NaN coordinates (T:402-460)."""
    seq = synth.make_sequence(2, 640, 480, 0.3, n_blobs=20000, seed=11)
    a, b = seq["images"]
    pts = H.shi_tomasi(O, "orc", a, 150, 0.01, 8)
    edge = np.array([[0.2, 0.3], [639.5, 479.5], [-3.0, 10.0], [5.0, -2.5], [638.9, 100.0], [100.0, 478.7], [1e12, 5.0], [np.nan, 7.0],
                     [320.0, 240.0], [-40.0, -40.0], [700.0, 500.0], [15.5, 15.5], [16.0, 464.0]])
    pts = np.ascontiguousarray(np.concatenate([pts, edge]))
    pa, pb = ctx.pyramid(a, 3), ctx.pyramid(b, 3)
    efwd, eback, ekeep = H.klt_track(O, "orc", a, b, 3, radius, 6, pts, 1.0)
    for sums in ("mfma", "valu"):
        for pipe_ in ("0", "1"):
            monkeypatch.setenv("SFMX_KLT_SUMS", sums)
            monkeypatch.setenv("SFMX_KLT_PIPE", pipe_)
            fwd, back, keep, steps = ctx.klt_track(pa, pb, pts, 3, radius, 6, 1.0)
            H.assert_bits_equal(fwd, efwd, f"fwd sums={sums} pipe={pipe_}", nan_equal=True)
            H.assert_bits_equal(back, eback, f"back sums={sums} pipe={pipe_}", nan_equal=True)
            assert np.array_equal(keep, ekeep), (sums, pipe_)


def test_hypothesis_loop_variants_identical(ctx, golden, monkeypatch):
    """k_hypotheses: this round's rotation loop (two-pass key maximum, stop / near-tie tests from the reduced key) and the round-2
    loop (SFMX_RANSAC_HYP=legacy) produce the same hypotheses, conditioning estimates, flags and counts, bit for bit."""
    xi, xj = golden["tv_xi"], golden["tv_xj"]
    idx8 = H.uniform_draws(O, "orc", 777, len(xi), 8 * 2500).reshape(2500, 8)
    out = {}
    for var in ("legacy", "lean"):
        monkeypatch.setenv("SFMX_RANSAC_HYP", var)
        out[var] = ctx.ransac_score_ex(xi, xj, idx8, 1e-3)
    for k in ("E", "cond", "flags", "counts", "lo", "hi"):
        H.assert_bits_equal(np.ascontiguousarray(out["lean"][k]), np.ascontiguousarray(out["legacy"][k]), k)
    assert (out["lean"]["best_iter"], out["lean"]["best_count"]) == (out["legacy"]["best_iter"], out["legacy"]["best_count"])


def test_klt_empty_and_radius_limits(ctx, golden):
    pa, pb = ctx.pyramid(golden["klt_a"], 3), ctx.pyramid(golden["klt_b"], 3)
    fwd, back, keep, steps = ctx.klt_track(pa, pb, np.zeros((0, 2)))
    assert fwd.shape == (0, 2) and steps == 0
    pts = golden["klt_pts"][:16]
    fwd, back, keep, _ = ctx.klt_track(pa, pb, pts, 3, 7, 3, 1.0)
    efwd, eback, ekeep = H.klt_track(O, "orc", golden["klt_a"], golden["klt_b"], 3, 7, 3, pts, 1.0)
    H.assert_bits_equal(fwd, efwd, "r=7 fwd")
    H.assert_bits_equal(back, eback, "r=7 back")
    with pytest.raises(capi.SfmxError):
        ctx.klt_track(pa, pb, pts, 3, 8, 3, 1.0)


def _check_certified(res, Eref, cref, what):
    """Contract of sfmx_ransac_score_ex (include/sfmx.h) against the reference's hypotheses / counts of EVERY iteration:
    exact rows are the reference's E bit for bit, every count equals the reference's, and it lies inside [lo, hi]."""
    ex = res["flags"].astype(bool)
    H.assert_bits_equal(res["E"][ex], Eref[ex], f"{what}: exact (host libm) hypotheses")
    assert np.array_equal(res["counts"], cref), (what, np.nonzero(res["counts"] != cref)[0][:10])
    assert np.all(res["lo"] <= cref) and np.all(cref <= res["hi"]), what
    assert np.array_equal(res["lo"][ex], cref[ex]) and np.array_equal(res["hi"][ex], cref[ex]), what
    best = int(np.argmax(cref))  # first maximum == lowest iteration on ties (T:673)
    assert (res["best_iter"], res["best_count"]) == (best, int(cref[best])), what
    # device hypotheses: the distance to the reference's E that the count bounds are built on (RANSAC_DEV_EPS / cond in
    # csrc/hip/ransac.hip) must hold with a factor 10 to spare
    scale = np.abs(Eref).max(axis=(1, 2))
    err = np.abs(res["E"] - Eref).max(axis=(1, 2)) / scale
    assert np.all(err[~ex] * res["cond"][~ex] < 1e-17), (what, (err[~ex] * res["cond"][~ex]).max())
    assert np.all(res["cond"][~ex] >= 1e-13) and np.all(np.isinf(res["cond"][ex])), what
    return ex


def test_ransac_counts_and_mask(ctx, golden):
    xi, xj, idx8 = golden["tv_xi"], golden["tv_xj"], golden["tv_idx8"]
    res = ctx.ransac_score_ex(xi, xj, idx8, 1e-3)
    # reference hypotheses (libm Jacobi) and their counts, every octet (incl. the repeated-index ones)
    Eref = golden["tv_E"]
    cref = np.zeros(len(idx8), np.int32)
    O.call("orc_ransac_counts", None, H.f64(xi), H.f64(xj), len(xi), H.f64(Eref), len(idx8), 1e-3, cref)
    ex = _check_certified(res, Eref, cref, "golden tv")
    deg = np.array([len(set(r)) < 8 for r in idx8])
    assert deg.any() and np.all(ex[deg]), "repeated-index octets must be scored with the exact host hypothesis"
    counts, bi, bc, _ = ctx.ransac_score(xi, xj, idx8, 1e-3)  # the plain entry point is the same computation
    assert np.array_equal(counts, cref) and (bi, bc) == (res["best_iter"], res["best_count"])
    # exact mask for a given E (bit-exact Sampson arithmetic)
    for k in (0, int(np.argmax(cref))):
        mask, cnt = ctx.sampson_mask(xi, xj, Eref[k], 1e-3)
        exp = np.array([H.sampson(O, "orc", Eref[k], xi[i], xj[i]) < 1e-3 for i in range(len(xi))])
        assert np.array_equal(mask.astype(bool), exp) and cnt == exp.sum()


def test_ransac_full_call_every_iteration(ctx, golden):
    """2500 hypotheses (BASELINE call shape, T:1739): every iteration's count equals the oracle's, the winner is the
    iteration the oracle's find_E_ransac picks, and the host seam returns its inliers / R / t."""
    K, pi, pj = golden["tv_K"], golden["tv_pi"], golden["tv_pj"]
    xi, xj = golden["tv_xi"], golden["tv_xj"]
    n = len(pi)
    idx8 = H.uniform_draws(O, "orc", 12345, n, 8 * 2500).reshape(2500, 8)
    Eref = np.array([H.eight_point(O, "orc", xi, xj, d) for d in idx8])
    cref = np.zeros(2500, np.int32)
    O.call("orc_ransac_counts", None, H.f64(xi), H.f64(xj), n, H.f64(Eref), 2500, 1e-3, cref)
    res = ctx.ransac_score_ex(xi, xj, idx8, 1e-3)
    _check_certified(res, Eref, cref, "2500 iterations")
    r = H.find_E_ransac(O, "orc", K, pi, pj, 2500, 1e-3, 60)
    assert r["ok"] == 1 and res["best_iter"] == r["best_iter"] and res["best_count"] == len(r["inliers"])
    mask, cnt = ctx.sampson_mask(xi, xj, r["E"], 1e-3)
    assert np.array_equal(np.nonzero(mask)[0], r["inliers"])
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    got = pipe.find_E_ransac(ctx, K, pi, pj, 2500, 1e-3, 60)
    assert got["ok"] == 1 and got["best_iter"] == r["best_iter"] and np.array_equal(got["inliers"], r["inliers"])
    H.assert_bits_equal(got["R"], r["R"], "R")
    H.assert_bits_equal(got["t"], r["t"], "t")


def test_find_E_ransac_seam_vs_reference_goldens(ctx, golden):
    """The find_E_ransac seam against what the compiled reference returned (rs_* in hotpath.npz)."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    for iters, thr, mi in golden["rs_cases"]:
        tag = f"{int(iters)}_{int(mi)}"
        got = pipe.find_E_ransac(ctx, golden["tv_K"], golden["tv_pi"], golden["tv_pj"], int(iters), float(thr), int(mi))
        assert got["ok"] == int(golden[f"rs_ok_{tag}"][0]), tag
        if got["ok"]:
            assert np.array_equal(got["inliers"], golden[f"rs_inl_{tag}"]), tag
            H.assert_bits_equal(got["R"], golden[f"rs_R_{tag}"], "R " + tag)
            H.assert_bits_equal(got["t"], golden[f"rs_t_{tag}"], "t " + tag)
    assert pipe.find_E_ransac(ctx, golden["tv_K"], golden["tv_pi"][:7], golden["tv_pj"][:7], 10, 1e-3, 1)["ok"] == 0  # T:648


@pytest.mark.parametrize("kind", ["deg_wins", "deg_ties", "clean_wins"])
def test_ransac_degenerate_octets(ctx, kind):
    """tests/golden/ransac_degenerate.npz, generated from the compiled reference: a repeated-index octet is the
    reference's winner / ties the maximum earlier than a clean one / ties it later (sampling with replacement, T:665;
    strict '>' at T:673).  Counts of every iteration, winner, inliers, R and t must be the reference's."""
    g = np.load(os.path.join(H.GOLDEN, "ransac_degenerate.npz"))
    iters, thr, mi = g[f"{kind}_args"]
    xi, xj, idx8 = g[f"{kind}_xi"], g[f"{kind}_xj"], g[f"{kind}_idx8"]
    res = ctx.ransac_score_ex(xi, xj, idx8, float(thr))
    ex = _check_certified(res, g[f"{kind}_E"], g[f"{kind}_counts"], kind)
    assert np.all(ex[g[f"{kind}_deg"]])
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    got = pipe.find_E_ransac(ctx, g[f"{kind}_K"], g[f"{kind}_pi"], g[f"{kind}_pj"], int(iters), float(thr), int(mi))
    assert got["ok"] == 1 and got["best_iter"] == int(g[f"{kind}_best"][0])
    assert np.array_equal(got["inliers"], g[f"{kind}_inl"])
    H.assert_bits_equal(got["R"], g[f"{kind}_R"], "R")
    H.assert_bits_equal(got["t"], g[f"{kind}_t"], "t")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_find_E_ransac_as_n_ranks_is_the_single_gpu_result(ctx, golden, world):
    """Hypothesis sharding is exact: each virtual rank's winner of its own iteration range (ransac_local), merged the way the
    two all-reduce(max) merge them (packed (count, ~iteration) key, winner's E as raw bits), must give every rank -- the one
    that holds the winner and one that recomputes the mask from the E bits -- the reference's inliers, R and t.  Includes the
    fixtures where a repeated-index octet wins or ties (ties across ranks resolve to the lowest iteration, T:673)."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    for iters, thr, mi in golden["rs_cases"]:
        tag = f"{int(iters)}_{int(mi)}"
        for as_rank in (0, world - 1):
            got = pipe.find_E_ransac_world(ctx, golden["tv_K"], golden["tv_pi"], golden["tv_pj"], int(iters), float(thr), int(mi), world, as_rank)
            assert got["ok"] == int(golden[f"rs_ok_{tag}"][0]), (tag, as_rank)
            if got["ok"]:
                assert np.array_equal(got["inliers"], golden[f"rs_inl_{tag}"]), (tag, as_rank)
                H.assert_bits_equal(got["R"], golden[f"rs_R_{tag}"], f"R {tag} rank {as_rank}")
                H.assert_bits_equal(got["t"], golden[f"rs_t_{tag}"], f"t {tag} rank {as_rank}")
    g = np.load(os.path.join(H.GOLDEN, "ransac_degenerate.npz"))
    for kind in ("deg_wins", "deg_ties", "clean_wins"):
        iters, thr, mi = g[f"{kind}_args"]
        for as_rank in range(world):
            got = pipe.find_E_ransac_world(ctx, g[f"{kind}_K"], g[f"{kind}_pi"], g[f"{kind}_pj"], int(iters), float(thr), int(mi), world, as_rank)
            assert got["ok"] == 1 and got["best_iter"] == int(g[f"{kind}_best"][0]), (kind, as_rank)
            assert np.array_equal(got["inliers"], g[f"{kind}_inl"])
            H.assert_bits_equal(got["R"], g[f"{kind}_R"], "R")
            H.assert_bits_equal(got["t"], g[f"{kind}_t"], "t")


@pytest.mark.parametrize("case", [(2, 30), (6, 80), (10, 120)])
def test_ba_build_and_step(ctx, golden, case):
    W, P = case
    tag = f"{W}_{P}"
    K, poses, X = golden[f"ba_K_{tag}"], golden[f"ba_poses_{tag}"], golden[f"ba_X_{tag}"]
    optr, okf, ouv = golden[f"ba_optr_{tag}"], golden[f"ba_okf_{tag}"], golden[f"ba_ouv_{tag}"]
    nk = poses.shape[0]
    w0 = nk - W
    # window-local observation lists in the reference's map iteration order (host-side bookkeeping)
    order = H.map_iteration_order(O, "orc", P)
    Xs, ptr, li, uv = [], [0], [], []
    for pid in order:
        sel = [(okf[o] - w0, ouv[o]) for o in range(optr[pid], optr[pid + 1]) if okf[o] >= w0]
        if len(sel) < 2:
            continue
        Xs.append(X[pid])
        for a, b in sel:
            li.append(a)
            uv.append(b)
        ptr.append(len(li))
    Xs, ptr, li, uv = np.array(Xs), np.array(ptr, np.int32), np.array(li, np.int32), np.array(uv)
    pw = np.zeros((W, 12))
    for k in range(W):
        Rcw = poses[w0 + k, :9].reshape(3, 3)
        C = poses[w0 + k, 9:]
        # inv_wc (T:163-167): Rwc = R^T, twc = -(Rwc * t) with the reference's row-wise products
        Rwc = Rcw.T.copy()
        t = np.array([-(Rwc[r, 0] * C[0] + Rwc[r, 1] * C[1] + Rwc[r, 2] * C[2]) for r in range(3)])
        pw[k, :9], pw[k, 9:] = Rwc.ravel(), t
    prob = ctx.ba_problem(W, Xs, ptr, li, uv)
    S, b = prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, True)
    eS, eb = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
    O.call("orc_ba_build", None, H.f64(pw), W, H.f64(Xs), len(Xs), ptr, li, H.f64(uv), float(K[0, 0]), float(K[1, 1]),
           float(K[0, 2]), float(K[1, 2]), 3.0, 1e-3, 1, eS, eb)
    H.assert_bits_equal(S, eS, "S")
    H.assert_bits_equal(b, eb, "b")
    rc, dx = prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    erc, ex = H.solve_gauss(O, "orc", eS, eb)
    assert rc == 0 and erc == 0
    H.assert_bits_equal(dx, ex, "dx")
    prob.close()


def test_ba_full_size_c4(ctx):
    """BASELINE config C4: W=10 poses, P=50 000 points seen in every pose (500 k residuals), N(0,0.5 px)
    noise.  S, b and dx of one iteration must equal the oracle bit for bit at the full size, not only on small cases."""
    W, P = 10, 50000
    rng = np.random.default_rng(1)
    pw = np.zeros((W, 12))
    for k in range(W):
        R, t = synth.ring_pose(2.0 * k)
        pw[k, :9], pw[k, 9:] = R.ravel(), t
    K = synth.K_TEMPLE
    X = rng.normal(size=(P, 3)) * 0.08
    ptr = np.arange(0, (P + 1) * W, W, dtype=np.int32)
    li = np.tile(np.arange(W, dtype=np.int32), P)
    Xc = np.einsum("kij,pj->pki", pw[:, :9].reshape(W, 3, 3), X) + pw[None, :, 9:]
    uv = np.stack([K[0, 0] * Xc[..., 0] / Xc[..., 2] + K[0, 2], K[1, 1] * Xc[..., 1] / Xc[..., 2] + K[1, 2]], -1)
    uv = np.ascontiguousarray((uv + rng.normal(size=uv.shape) * 0.5).reshape(P * W, 2))
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    S, b = prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, True)
    eS, eb = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
    O.call("orc_ba_build", None, H.f64(pw), W, H.f64(X), P, ptr, li, H.f64(uv), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
           float(K[1, 2]), 3.0, 1e-3, 1, eS, eb)
    H.assert_bits_equal(S, eS, "S (C4)")
    H.assert_bits_equal(b, eb, "b (C4)")
    rc, dx = prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    erc, ex = H.solve_gauss(O, "orc", eS, eb)
    assert rc == 0 and erc == 0
    H.assert_bits_equal(dx, ex, "dx (C4)")
    prob.close()


def test_ba_edge_cases(ctx):
    """>16 observations (skipped, T:915), point behind a camera (T:933), singular Hpp (T:1012), duplicate pose."""
    rng = np.random.default_rng(4)
    W = 4
    pw = np.zeros((W, 12))
    for k in range(W):
        R, t = synth.ring_pose(3.0 * k)
        pw[k, :9], pw[k, 9:] = R.ravel(), t
    K = synth.K_TEMPLE
    X = rng.normal(size=(6, 3)) * 0.05
    X[1] = [0, 0, -5.0]  # behind every camera -> all residuals skipped -> Hpp = 0 -> inv3 fails
    ptr, li, uv = [0], [], []
    for p in range(6):
        ks = list(range(W)) if p != 2 else [0, 1, 1, 2]  # duplicate local pose index for p == 2
        if p == 3:
            ks = [k % W for k in range(17)]  # 17 observations -> skipped
        for k in ks:
            Xc = pw[k, :9].reshape(3, 3) @ X[p] + pw[k, 9:]
            li.append(k)
            uv.append([K[0, 0] * Xc[0] / Xc[2] + K[0, 2] + rng.normal(), K[1, 1] * Xc[1] / Xc[2] + K[1, 2] + rng.normal() * (20 if p == 4 else 1)])
        ptr.append(len(li))
    ptr, li, uv = np.array(ptr, np.int32), np.array(li, np.int32), np.array(uv)
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    for damp in (True, False):
        S, b = prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, damp)
        eS, eb = np.zeros((24, 24)), np.zeros(24)
        O.call("orc_ba_build", None, H.f64(pw), W, H.f64(X), 6, ptr, li, H.f64(uv), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
               float(K[1, 2]), 3.0, 1e-3, int(damp), eS, eb)
        H.assert_bits_equal(S, eS, "S")
        H.assert_bits_equal(b, eb, "b")


def test_solve_dense(ctx, golden):
    for n in (6, 36, 60, 7):
        rc, x = ctx.solve_dense(golden[f"sg_A_{n}"], golden[f"sg_b_{n}"])
        assert rc == 0
        H.assert_bits_equal(x, golden[f"sg_x_{n}"], f"solve n={n}")
    rc, _ = ctx.solve_dense(golden["sg_A_sing"], np.ones(5))
    assert rc == capi.SFMX_ERR_SINGULAR
    rng = np.random.default_rng(9)
    for n in (1, 2, 33, 96, 128):
        A = rng.normal(size=(n, n))
        b = rng.normal(size=n)
        if n == 33:
            A[5, :] *= 1e-20  # exercises the |f| < 1e-18 skip
        rc, x = ctx.solve_dense(A, b)
        erc, ex = H.solve_gauss(O, "orc", A, b)
        assert (rc != 0) == (erc != 0)
        if rc == 0:
            H.assert_bits_equal(x, ex, f"random n={n}")
    A = rng.normal(size=(9, 9))
    A[4, 4] = np.nan  # NaN propagates instead of throwing, as in the reference
    rc, x = ctx.solve_dense(A, np.ones(9))
    erc, ex = H.solve_gauss(O, "orc", A, np.ones(9))
    assert rc == erc == 0
    H.assert_bits_equal(x, ex, "nan solve", nan_equal=True)


@pytest.mark.parametrize("n", [36, 60])
def test_solve_register_kernels_edge_cases(ctx, n):
    """k_solve_regs<36/60> (the BA windows of 6 and 10 poses): random systems, pivot ties (first position wins), the
    |f| < 1e-18 skip, NaN on and off the diagonal, zero columns / singular systems, all against the oracle bit for bit."""
    rng = np.random.default_rng(100 + n)
    cases = []
    for t in range(6):
        cases.append((f"random {t}", rng.normal(size=(n, n)), rng.normal(size=n)))
    M = rng.normal(size=(n, n))
    cases.append(("spd", M @ M.T + 1e-3 * np.eye(n), rng.normal(size=n)))
    A = rng.integers(-3, 4, size=(n, n)).astype(np.float64)  # many equal |values| per column: ties at most steps
    cases.append(("integer ties", A + 0.0, rng.integers(-5, 6, size=n).astype(np.float64)))
    A = rng.normal(size=(n, n))
    A[:, 0] = np.where(np.arange(n) % 2 == 0, 2.5, -2.5)  # every row ties in the first column, signs differ
    cases.append(("tie column 0", A, rng.normal(size=n)))
    A = rng.normal(size=(n, n))
    A[5, :] *= 1e-20
    A[n - 2, :] *= 1e-19
    cases.append(("tiny multipliers", A, rng.normal(size=n)))
    A = rng.normal(size=(n, n))
    A[7, 3] = np.nan
    cases.append(("nan off the diagonal", A, np.ones(n)))
    A = rng.normal(size=(n, n))
    A[0, 0] = np.nan
    cases.append(("nan on the first diagonal element", A, np.ones(n)))
    A = rng.normal(size=(n, n))
    A[:, 4] = 0.0
    cases.append(("zero column", A, np.ones(n)))
    A = rng.normal(size=(n, n))
    A[n - 1, :] = A[0, :]
    cases.append(("duplicate row", A, np.ones(n)))
    cases.append(("identity with signed zeros", np.eye(n) * -1.0 + 0.0 * rng.normal(size=(n, n)), -np.ones(n)))
    for name, A, b in cases:
        rc, x = ctx.solve_dense(A, b)
        erc, ex = H.solve_gauss(O, "orc", A, b)
        assert (rc != 0) == (erc != 0), name
        if rc == 0:
            H.assert_bits_equal(x, ex, f"n={n} {name}", nan_equal=True)


def test_solve_dense_blocked_sizes(ctx):
    """n > 64 takes the blocked multi-workgroup elimination (pose graphs: 3 unknowns per keyframe).  Bit-exact against the
    oracle's solve_gauss on sizes around the block edges, on a pose-graph-shaped system, with tied pivots, skipped
    multipliers, NaN and a singular matrix deep inside."""
    rng = np.random.default_rng(21)
    for n in (65, 96, 97, 141, 300, 513):
        A = rng.normal(size=(n, n))
        b = rng.normal(size=n)
        if n == 97:
            A[40, :] *= 1e-20          # |f| < 1e-18 skip in the panel, in the block rows and in the trailing update
            A[:, 70] *= 1e-21
        if n == 141:
            A[100] = A[20]             # duplicate rows: tied pivot candidates, later an exactly singular step
        if n == 300:
            A = np.round(A * 4) / 4    # many exactly equal |a_ik|: first-maximum rule
        rc, x = ctx.solve_dense(A, b)
        erc, ex = H.solve_gauss(O, "orc", A, b)
        assert (rc != 0) == (erc != 0), n
        if rc == 0:
            H.assert_bits_equal(x, ex, f"blocked n={n}")
    # pose-graph shape: weighted graph Laplacian (x) I3 plus a gauge term, 100 keyframes -> 300 unknowns
    N = 100
    L = np.zeros((3 * N, 3 * N))
    for a in range(N - 1):
        for c, w in ((a + 1, 400.0 + a), (min(N - 1, a + 7), 90.0)):
            for d in range(3):
                L[3 * a + d, 3 * a + d] += w; L[3 * c + d, 3 * c + d] += w
                L[3 * a + d, 3 * c + d] -= w; L[3 * c + d, 3 * a + d] -= w
    L[:3, :3] += np.eye(3) * 1e9
    g = rng.normal(size=3 * N)
    rc, x = ctx.solve_dense(L, g)
    erc, ex = H.solve_gauss(O, "orc", L, g)
    assert rc == erc == 0
    H.assert_bits_equal(x, ex, "pose-graph system")
    A = rng.normal(size=(130, 130))
    A[77, 3] = np.nan
    rc, x = ctx.solve_dense(A, np.ones(130))
    erc, ex = H.solve_gauss(O, "orc", A, np.ones(130))
    assert rc == erc
    if rc == 0:
        H.assert_bits_equal(x, ex, "nan blocked", nan_equal=True)
    A = rng.normal(size=(200, 200))
    A[:, 150] = A[:, 10] * 2.0     # rank deficient: the pivot of some step deep inside falls below 1e-15 (or rounding keeps it alive -- same verdict either way)
    rc, x = ctx.solve_dense(A, np.ones(200))
    erc, ex = H.solve_gauss(O, "orc", A, np.ones(200))
    assert (rc != 0) == (erc != 0)
    if rc == 0:
        H.assert_bits_equal(x, ex, "rank-deficient blocked", nan_equal=True)


def test_solve_dense_has_no_size_cliff(ctx, monkeypatch):
    """n > 6400: the back-substitution's x and staged rows no longer fit in LDS and move to a global scratch buffer -- same
    subtraction chain (dense.hpp:86-91), still bit-exact.  The global path at small n (forced), then for real at n = 6600
    on a banded system (the oracle skips exact-zero multipliers, dense.hpp:80, so it finishes in seconds; the device
    code runs its full dense schedule either way)."""
    rng = np.random.default_rng(31)
    monkeypatch.setenv("SFMX_BACKSUB_LDS_MAX_N", "64")
    for n in (97, 300, 513):
        A = rng.normal(size=(n, n))
        b = rng.normal(size=n)
        rc, x = ctx.solve_dense(A, b)
        erc, ex = H.solve_gauss(O, "orc", A, b)
        assert rc == erc == 0
        H.assert_bits_equal(x, ex, f"global back-substitution n={n}")
    monkeypatch.delenv("SFMX_BACKSUB_LDS_MAX_N")
    n, bw = 6600, 24
    A = np.zeros((n, n))
    for d in range(-bw, bw + 1):
        v = rng.normal(size=n - abs(d))
        A[np.arange(max(0, -d), min(n, n - d)), np.arange(max(0, d), min(n, n + d))] = v
    A[np.arange(n), np.arange(n)] += 3.0 * bw ** 0.5
    b = rng.normal(size=n)
    rc, x = ctx.solve_dense(A, b)
    erc, ex = H.solve_gauss(O, "orc", A, b)
    assert rc == erc == 0
    H.assert_bits_equal(x, ex, "n = 6600 > LDS limit")


def _pose_graph(N, loops, seed, break_at=None):
    """keyframes on a noisy ring, odometry edges i -> i+1 and `loops` loop edges; break_at: drop that odometry edge"""
    rng = np.random.default_rng(seed)
    Rs = np.zeros((N, 9))
    C = np.zeros((N, 3))
    for k in range(N):
        R, t = synth.ring_pose(0.05 * k)
        Rs[k] = R.T.ravel()                     # camera -> world
        C[k] = -R.T @ t + rng.normal(size=3) * 1e-3
    ei, ej, eR, et, lp = [], [], [], [], []
    pairs = [(i, i + 1, 0) for i in range(N - 1) if i != break_at]
    for _ in range(loops):
        a = int(rng.integers(0, N - 8))
        pairs.append((a, int(rng.integers(a + 6, N)), 1))
    for i, j, l in pairs:
        Rw_i, Rw_j = Rs[i].reshape(3, 3), Rs[j].reshape(3, 3)
        R_ji = Rw_j.T @ Rw_i                    # i -> j
        t_ji = Rw_j.T @ (C[i] - C[j]) + rng.normal(size=3) * 1e-3
        t_ji /= np.linalg.norm(t_ji)
        ei.append(i); ej.append(j); eR.append(R_ji.ravel()); et.append(t_ji); lp.append(l)
    return Rs, C, np.array(ei, np.int32), np.array(ej, np.int32), np.array(eR), np.array(et), np.array(lp, np.int32)


@pytest.mark.parametrize("N", [300, 1000, 3000])
def test_posegraph_structured_solver_vs_dense_oracle(ctx, N, monkeypatch):
    """posegraph_optimize_centers (T:1131-1197) with the structured FP64-MFMA solver (tolerance mode, the product path above
    3N = 6400 unknowns) against the oracle's dense 3N x 3N solve_gauss: n = 900, 3000, 9000 unknowns, centres within
    1e-9 relative; a graph with a keyframe cut off from node 0 is skipped by both."""
    pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
    Rs, C, ei, ej, eR, et, lp = _pose_graph(N, max(4, N // 40), 2)
    ok_o, Co = H.posegraph(O, "orc", Rs, C, ei, ej, eR, et, lp)
    monkeypatch.setenv("SFMX_POSEGRAPH_SOLVER", "structured")
    ok_s, Cs = pipe.posegraph(ctx, Rs, C, ei, ej, eR, et, lp)
    assert ok_o == 1 and ok_s == 1
    scale = np.abs(Co - C).max()
    assert scale > 1e-6  # the solve must actually move the centres
    assert np.abs(Cs - Co).max() <= 1e-9 * max(scale, np.abs(Co).max()), np.abs(Cs - Co).max()
    if N <= 1000:  # the bit-exact dense path on the same input
        monkeypatch.setenv("SFMX_POSEGRAPH_SOLVER", "dense")
        ok_d, Cd = pipe.posegraph(ctx, Rs, C, ei, ej, eR, et, lp)
        assert ok_d == 1
        H.assert_bits_equal(Cd, Co, "dense pose-graph path")
        monkeypatch.setenv("SFMX_POSEGRAPH_SOLVER", "structured")
    Rs, C, ei, ej, eR, et, lp = _pose_graph(N, 0, 3, break_at=N // 2)  # two components: singular
    ok_o, _ = H.posegraph(O, "orc", Rs, C, ei, ej, eR, et, lp)
    ok_s, Cs = pipe.posegraph(ctx, Rs, C, ei, ej, eR, et, lp)
    assert ok_o == 0 and ok_s == 0 and np.array_equal(Cs, C)


def _c4_like_problem(W, P, seed):
    rng = np.random.default_rng(seed)
    pw = np.zeros((W, 12))
    for k in range(W):
        R, t = synth.ring_pose(3.0 * k)
        pw[k, :9], pw[k, 9:] = R.ravel(), t
    K = synth.K_TEMPLE
    X = rng.normal(size=(P, 3)) * 0.05
    ptr = np.arange(0, (P + 1) * W, W, dtype=np.int32)
    li = np.tile(np.arange(W, dtype=np.int32), P)
    Xc = np.einsum("kij,pj->pki", pw[:, :9].reshape(W, 3, 3), X) + pw[None, :, 9:]
    uv = np.stack([K[0, 0] * Xc[..., 0] / Xc[..., 2] + K[0, 2], K[1, 1] * Xc[..., 1] / Xc[..., 2] + K[1, 2]], -1)
    uv = np.ascontiguousarray((uv + rng.normal(size=uv.shape)).reshape(P * W, 2))
    return pw, K, X, ptr, li, uv


@pytest.mark.parametrize("shape", [(6, 600), (10, 3000), (6, 40)])
def test_ba_step_host_solve_equals_device_solve(ctx, shape, monkeypatch):
    """sfmx_ba_step solves the window's 36 / 60 unknowns on the host core that polls for the result (csrc/hip/solve_host.cpp,
    the default) or in the last workgroup of the reduction (SFMX_BA_SOLVE=device): same S | b, the same elimination order
    (dense.hpp:54-93), so dx and the status must agree bit for bit -- with each other and with the oracle -- also where the
    system is singular (a point cloud behind the cameras: every residual skipped, T:933) and where a pose is NaN."""
    W, P = shape
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 21)
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    a = (K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    behind = pw.copy()
    behind[:, 11] -= 10.0   # every Xc.z <= 1e-6: S = lambda I + gauge, b = 0: solvable, dx = 0
    nanpose = pw.copy()
    nanpose[2, 4] = np.nan
    for tag, poses in (("regular", pw), ("behind", behind), ("nan", nanpose)):
        monkeypatch.delenv("SFMX_BA_SOLVE", raising=False)
        rc_h, dx_h = prob.step(poses, *a)
        monkeypatch.setenv("SFMX_BA_SOLVE", "device")
        rc_d, dx_d = prob.step(poses, *a)
        monkeypatch.delenv("SFMX_BA_SOLVE", raising=False)
        S, b = prob.build(poses, *a, True)
        erc, ex = H.solve_gauss(O, "orc", S, b)
        assert rc_h == rc_d == (0 if erc == 0 else capi.SFMX_ERR_SINGULAR), (tag, rc_h, rc_d, erc)
        if erc == 0:
            H.assert_bits_equal(dx_h, ex, f"host solve vs oracle ({tag})", nan_equal=True)
            H.assert_bits_equal(dx_d, ex, f"device solve vs oracle ({tag})", nan_equal=True)
    prob.close()


@pytest.mark.parametrize("shape", [(6, 600), (6, 17), (10, 300), (3, 33), (2, 4000)])
def test_ba_build_window_shapes_equal_oracle(ctx, shape):
    """sfmx_ba_build on window-sized problems of several shapes (2 to 10 poses, 17 to 4 000 points) against the oracle's build:
    S and b bit for bit, with and without damping / gauge."""
    W, P = shape
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 41)
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    for damp in (True, False):
        S, b = prob.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, damp)
        eS, eb = np.zeros((6 * W, 6 * W)), np.zeros(6 * W)
        O.call("orc_ba_build", None, H.f64(pw), W, H.f64(X), P, H.i32(ptr), H.i32(li), H.f64(uv), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
               float(K[1, 2]), 3.0, 1e-3, int(damp), eS, eb)
        H.assert_bits_equal(S, eS, f"S W={W} P={P} damp={damp}")
        H.assert_bits_equal(b, eb, f"b W={W} P={P} damp={damp}")
    prob.close()


@pytest.mark.parametrize("P", [600, 37, 3])
def test_ba_resident_job_equals_plain_steps(ctx, P, monkeypatch):
    """sfmx_ba_begin / sfmx_ba_step x n / sfmx_ba_end: the resident kernel of a job (one launch, iterations driven through a
    command word in pinned memory) must return, step for step, the dx of the plain two-launches-per-step path -- on a chain of
    iterations whose poses depend on the previous dx, on a job that is ended early, and on a job that is longer than the
    kernel was started for (the surplus steps take the plain path)."""
    W = 6
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 31)
    a = (K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    prob = ctx.ba_problem(W, X, ptr, li, uv)

    def chain(n, resident, iters=None, end_after=None):
        poses = pw.copy()
        out = []
        if resident:
            prob.begin(iters if iters is not None else n, *a)
        for k in range(n):
            if end_after is not None and k == end_after:
                prob.end()
            rc, dx = prob.step(poses, *a)
            assert rc == 0
            out.append(dx.copy())
            poses[1:, 9:] += 1e-3 * dx.reshape(W, 6)[1:, 3:]  # feed the step back (any deterministic function of dx)
        if resident:
            prob.end()
        return out

    monkeypatch.setenv("SFMX_BA_RESIDENT", "0")
    ref = chain(5, False)
    monkeypatch.setenv("SFMX_BA_RESIDENT", "1")  # opt-in: measured slower inside the pipeline
    for tag, got in (("5 of 5", chain(5, True)), ("ended after 2", chain(5, True, end_after=2)), ("7 steps on a 5-iteration job", chain(5, True, iters=3)),
                     ("again", chain(5, True))):
        for k in range(5):
            H.assert_bits_equal(got[k], ref[k], f"resident job ({tag}) step {k}")
    prob.close()


def test_ba_sharded_step_single_rank_equals_fused_step(ctx):
    """sfmx_ba_step_sharded (partial build -> [all-reduce] -> damp/gauge kernel -> solve) with one rank must equal the fused
    sfmx_ba_step bit for bit: same sums, no collective reordering with world size 1 -- with no communicator and with a
    world-size-1 communicator object."""
    dist_mod = importlib.import_module(H.PKG_NAME + ".dist")
    W, P = 4, 50
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 8)
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    rc1, dx1 = prob.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    rc2, dx2 = dist_mod.ba_step_sharded(ctx, prob, pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, None)
    comm1 = capi.Comm(0, None, 0, 1)
    rc3, dx3 = dist_mod.ba_step_sharded(ctx, prob, pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, comm1)
    assert rc1 == rc2 == rc3 == 0
    H.assert_bits_equal(dx1, dx2, "sharded (no comm) vs fused dx")
    H.assert_bits_equal(dx1, dx3, "sharded (world 1) vs fused dx")
    comm1.close()
    prob.close()


@pytest.mark.parametrize("shape", [(6, 600, 2), (6, 600, 8), (10, 20000, 8), (3, 7, 8)])
def test_ba_step_virtual_world(ctx, shape, monkeypatch):
    """SFMX_VIRTUAL_WORLD=N (test mode of the sharded BA steps): what N ranks compute, formed on one GPU.
    sfmx_ba_step_sharded_elements (disjoint slices of S | b, +0.0 elsewhere, summed in any order): dx bit for bit.
    sfmx_ba_step_sharded (shard sums over sfmx_shard_range's point ranges): `relay` (each shard continues the running sums of
    the one before) IS the reference's sequence, bit for bit; the all-reduce associations (rank order, reverse, ring,
    pairwise tree) regroup the addends: dx within 1e-9 per step."""
    W, P, world = shape
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 11)
    prob = ctx.ba_problem(W, X, ptr, li, uv)
    a = (pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    monkeypatch.delenv("SFMX_VIRTUAL_WORLD", raising=False)
    rc, dx_ref = prob.step(*a)
    assert rc == 0
    monkeypatch.setenv("SFMX_VIRTUAL_WORLD", str(world))
    for order in ("rank", "reverse", "ring", "tree"):  # element slices: zeros are all an all-reduce ever adds, whatever its order
        monkeypatch.setenv("SFMX_VIRTUAL_WORLD_ORDER", order)
        rc, dx = prob.step_sharded_elements(None, *a)
        assert rc == 0, order
        H.assert_bits_equal(dx, dx_ref, f"element-sharded, virtual world {world} {order}")
    for order in ("relay", "rank", "reverse", "ring", "tree"):
        monkeypatch.setenv("SFMX_VIRTUAL_WORLD_ORDER", order)
        rc, dx = prob.step_sharded(None, *a)
        assert rc == 0, order
        if order == "relay" or P < world:  # fewer points than ranks: no sharding at all
            H.assert_bits_equal(dx, dx_ref, f"virtual world {world} {order}")
        else:
            assert np.allclose(dx, dx_ref, rtol=0, atol=1e-9 * np.abs(dx_ref).max()), (order, np.abs(dx - dx_ref).max())
    prob.close()


def test_ba_point_shards_sum_to_the_fused_system(ctx):
    """The arithmetic of the N-rank mode on one GPU: the raw S | b of the contiguous point ranges sfmx_shard_range hands to
    ranks 0..N-1, summed in rank order (what the all-reduce does), must agree with the single-rank system to 1e-9
    relative (not bit for bit: the addends are grouped differently) and give a dx that close."""
    W, P = 10, 5000
    pw, K, X, ptr, li, uv = _c4_like_problem(W, P, 9)
    D = 6 * W
    full = ctx.ba_problem(W, X, ptr, li, uv)
    S_ref, b_ref = full.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, False)
    rc, dx_ref = full.step(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3)
    assert rc == 0
    for world in (2, 8):
        S, b = np.zeros((D, D)), np.zeros(D)
        covered = 0
        for rank in range(world):
            lo, hi = capi.shard_range(P, rank, world)
            assert lo == covered
            covered = hi
            o0, o1 = int(ptr[lo]), int(ptr[hi])
            shard = ctx.ba_problem(W, X[lo:hi], ptr[lo:hi + 1] - o0, li[o0:o1], uv[o0:o1])
            Sr, br = shard.build(pw, K[0, 0], K[1, 1], K[0, 2], K[1, 2], 3.0, 1e-3, False)
            S += Sr
            b += br
            shard.close()
        assert covered == P
        assert np.allclose(S, S_ref, rtol=0, atol=1e-9 * np.abs(S_ref).max())
        assert np.allclose(b, b_ref, rtol=0, atol=1e-9 * np.abs(b_ref).max())
        S[np.arange(D), np.arange(D)] += 1e-3
        S[np.arange(6), np.arange(6)] += 1e9
        b[:6] = 0.0
        rc, dx = ctx.solve_dense(S, b)
        assert rc == 0 and np.allclose(dx, dx_ref, rtol=0, atol=1e-7 * np.abs(dx_ref).max())
    full.close()
