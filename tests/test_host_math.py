"""CPU suite: the product's host-side math (libsfmx_host.so; libm-dependent pieces that stay on the
host by design) against the golden vectors from the reference build.  No device calls."""
import ctypes
import importlib
import os

import numpy as np

import helpers as H

pipe = importlib.import_module(H.PKG_NAME + ".pipeline")
L = H.CLib(pipe.HOST_LIB_PATH)


def test_host_rng_matches_libstdcxx(golden):
    for n in (8, 100, 517, 5000, 3):
        out = np.zeros(4096, np.int32)
        L.call("sfmx_host_uniform_draws", None, ctypes.c_uint(12345), n, 4096, out)
        assert np.array_equal(out, golden[f"rng_{n}"])


def test_host_eight_point_is_reference_exact(golden):
    xi, xj = golden["tv_xi"], golden["tv_xj"]
    for k, idx in enumerate(golden["tv_idx8"]):
        E = np.zeros((3, 3))
        L.call("sfmx_host_eight_point_E", None, H.f64(xi), H.f64(xj), H.i32(idx), E)
        H.assert_bits_equal(E, golden["tv_E"][k], f"E[{k}]")


def test_host_decompose_matches_find_E_ransac(golden):
    """winner E (oracle diagnostic, itself pinned to the reference) -> R,t must equal the reference's RelPose"""
    O = H.oracle()
    for iters, thr, mi in golden["rs_cases"]:
        tag = f"{int(iters)}_{int(mi)}"
        if not int(golden[f"rs_ok_{tag}"][0]):
            continue
        r = H.find_E_ransac(O, "orc", golden["tv_K"], golden["tv_pi"], golden["tv_pj"], int(iters), float(thr), int(mi))
        R, t = np.zeros((3, 3)), np.zeros(3)
        inl = H.i32(golden[f"rs_inl_{tag}"])
        L.call("sfmx_host_decompose_E", None, H.f64(r["E"]), H.f64(golden["tv_xi"]), H.f64(golden["tv_xj"]), inl, len(inl), R, t)
        H.assert_bits_equal(R, golden[f"rs_R_{tag}"], "R")
        H.assert_bits_equal(t, golden[f"rs_t_{tag}"], "t")


def test_host_triangulate_and_so3(golden):
    K = golden["tv_K"]
    for k, row in enumerate(golden["tri_in"]):
        X = np.zeros(3)
        L.call("sfmx_host_triangulate_dlt", None, H.f64(K).reshape(9), H.f64(row[0:9]), H.f64(row[9:12]), H.f64(row[12:21]),
               H.f64(row[21:24]), H.f64(row[24:26]), H.f64(row[26:28]), X)
        H.assert_bits_equal(X, golden["tri_out"][k], f"tri {k}")
    for k, w in enumerate(golden["so3_w"]):
        R, lg = np.zeros((3, 3)), np.zeros(3)
        L.call("sfmx_host_so3", None, H.f64(w), R, lg)
        H.assert_bits_equal(R, golden["so3_R"][k], "exp")
        H.assert_bits_equal(lg, golden["so3_log"][k], "log")


def test_hypot_restatement_matches_libm():
    """sfmx_math.h hypot_glibc (the function the device kernels use) vs the platform libm."""
    rng = np.random.default_rng(2)
    x = np.ldexp(rng.uniform(-1, 1, 200000), rng.integers(-40, 40, 200000))
    y = np.ldexp(rng.uniform(-1, 1, 200000), rng.integers(-40, 40, 200000))
    y[::5] = x[::5] * rng.uniform(-1, 1, x[::5].size)
    x[:1000] *= 1e-3
    fn = L.dll.sfmx_host_hypot
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_double, ctypes.c_double]
    got = np.array([fn(float(a), float(b)) for a, b in zip(x[:60000], y[:60000])])
    H.assert_bits_equal(got, np.hypot(x[:60000], y[:60000]), "hypot")
    for a, b in [(0.0, 0.0), (np.inf, 1.0), (3.0, 0.0), (0.0, -4.0), (1e300, 1e300), (1e-320, 1e-320), (5e-324, 0.0)]:
        assert fn(a, b) == np.hypot(a, b)
    assert np.isnan(fn(float("nan"), 1.0))


def test_introsort_replay_reproduces_std_sort_tie_order():
    """The partial replay of libstdc++'s introsort (host/introsort_replay.hpp) that decides the order of
    equal-score corner candidates: (1) the full replay equals the real std::sort element for element on
    tie-heavy inputs; (2) the selective replay puts every marked pair of equal keys in the same relative
    order as std::sort does."""
    rng = np.random.default_rng(12)
    fn = L.dll.sfmx_host_sort_order
    fn.restype = ctypes.c_int
    for n, levels in [(5, 2), (16, 3), (17, 3), (100, 7), (1000, 13), (20000, 500), (150000, 4000), (150000, 10**9), (3000, 1)]:
        for rep in range(3):
            scores = rng.integers(0, levels, n).astype(np.float64) * 0.25
            if rep == 1:
                scores = np.sort(scores)          # adversarial orders
            if rep == 2:
                scores = np.sort(scores)[::-1].copy()
            ref_ids = np.zeros(n, np.uint32)
            full_ids = np.zeros(n, np.uint32)
            assert fn(scores.ctypes.data_as(ctypes.c_void_p), None, n, 0, ref_ids.ctypes.data_as(ctypes.c_void_p)) == 1
            ok = fn(scores.ctypes.data_as(ctypes.c_void_p), None, n, 1, full_ids.ctypes.data_as(ctypes.c_void_p))
            if ok:  # the replay declines (returns 0) only when the real sort would switch to heapsort
                assert np.array_equal(ref_ids, full_ids), (n, levels, rep)
            marks = (rng.random(n) < min(1.0, 40.0 / n)).astype(np.uint8)
            sel_ids = np.zeros(n, np.uint32)
            ok = fn(scores.ctypes.data_as(ctypes.c_void_p), marks.ctypes.data_as(ctypes.c_void_p), n, 2, sel_ids.ctypes.data_as(ctypes.c_void_p))
            if ok:
                pos_ref = np.empty(n, np.int64); pos_ref[ref_ids] = np.arange(n)
                pos_sel = np.empty(n, np.int64); pos_sel[sel_ids] = np.arange(n)
                m = np.nonzero(marks)[0]
                for a in range(len(m)):
                    for b in range(a + 1, len(m)):
                        if scores[m[a]] == scores[m[b]]:
                            assert (pos_ref[m[a]] < pos_ref[m[b]]) == (pos_sel[m[a]] < pos_sel[m[b]]), (n, levels, rep)


def test_arena_map_iterates_like_default_unordered_map():
    """kf.obs / track_hist / map.pts live on a bump arena; their iteration order is part of the result (hazard H2) and
    must be what libstdc++'s default-allocator unordered_map gives for the same insert/erase history."""
    rng = np.random.default_rng(3)
    fn = L.dll.sfmx_host_map_order
    fn.restype = ctypes.c_int
    for n, span, erase in [(10, 20, 0.0), (1000, 5000, 0.0), (20000, 30000, 0.2), (50000, 10**9, 0.1), (5000, 40, 0.5)]:
        keys = rng.integers(0, span, n).astype(np.int32)
        ops = (rng.random(n) < erase).astype(np.uint8)
        a = np.full(n, -1, np.int32)
        b = np.full(n, -2, np.int32)
        k = fn(keys.ctypes.data_as(ctypes.c_void_p), ops.ctypes.data_as(ctypes.c_void_p), n, a.ctypes.data_as(ctypes.c_void_p),
               b.ctypes.data_as(ctypes.c_void_p))
        assert k >= 0
        assert np.array_equal(a[:k], b[:k]), (n, span, erase)


def test_sparse_mesh_matches_reference():
    """--export-geometry mesh (mesh.cpp): projection, shuffled grid thinning, Bowyer-Watson, long-edge filter.  Vertices
    (bitwise) and faces must be what the reference's build_mesh_from_sparse_points returned on the committed inputs
    (tests/golden/make_mesh_golden.py; incl. a regular lattice = cocircular points, < 50 points, all behind the camera)."""
    import os
    g = np.load(os.path.join(H.GOLDEN, "mesh.npz"))
    for name in [str(n) for n in g["names"]]:
        X = np.ascontiguousarray(g[name + "_X"])
        w, h, mp, gp, me = g[name + "_cfg"]
        cap = max(len(X), 1)
        v = np.zeros((cap, 3))
        f = np.zeros((cap * 4, 3), np.int32)
        nf = np.zeros(1, np.int32)
        nv = L.call("sfmx_host_sparse_mesh", int, H.f64(g[name + "_K"]), H.f64(g[name + "_pose"]), H.f64(X), len(X), int(w), int(h), int(mp),
                    int(gp), float(me), v, cap, f, cap * 4, nf)
        assert nv == len(g[name + "_v"]) and int(nf[0]) == len(g[name + "_f"]), name
        H.assert_bits_equal(v[:nv], g[name + "_v"], f"mesh vertices ({name})")
        assert np.array_equal(f[:int(nf[0])], g[name + "_f"]), name


# ------------------------------------------------------------------ file-format surface of the CLI (csrc/host/cli_io.hpp)
def _surface():
    import json
    return json.load(open(os.path.join(H.GOLDEN, "surface_cases.json")))


def test_read_pgm_matches_reference_reader(tmp_path):
    """PGM header comments, maxval != 255, short files, odd separators (cpp/include/pgm_io.hpp:24-54): size, pixels and
    error text of the product's reader vs what the compiled reference reader did (tests/golden/make_surface_golden.py)."""
    import base64
    import ctypes
    host = L.dll
    path = str(tmp_path / "f.pgm")
    for c in _surface()["pgm"]:
        with open(path, "wb") as f:
            f.write(base64.b64decode(c["data"]))
        w, h, cs = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_ulonglong(0)
        err = ctypes.create_string_buffer(512)
        rc = host.sfmx_host_read_pgm(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(cs), err, 512)
        msg = err.value.decode().replace(path, "<PATH>")
        if c["ub"]:  # the reference reads uninitialised ints here: it fails, with one of its two later error texts
            assert rc == 1 and msg in ("Only 8-bit PGM supported: <PATH>", "PGM read failed: <PATH>"), (c["name"], msg)
            continue
        assert rc == c["rc"], (c["name"], msg)
        if rc == 0:
            assert (w.value, h.value, cs.value) == (c["w"], c["h"], c["checksum"]), c["name"]
        else:
            assert msg == c["error"], c["name"]


def test_config_json_matches_reference_reader():
    """config.json lookups (cpp.* over common.*, llround, type mismatches, duplicate keys, escapes) and every parse-error
    text: the product's reader vs minijson + the getters of templering_sfm.cpp (T:65-106) compiled from the reference."""
    import base64
    import ctypes
    host = L.dll
    for c in _surface()["json"]:
        num = ctypes.c_double(0)
        buf = ctypes.create_string_buffer(512)
        rc = host.sfmx_host_config_lookup(c["text"].encode(), c["section"].encode(), c["key"].encode(), c["kind"], ctypes.byref(num), buf, 512)
        assert rc == c["rc"], (c["name"], rc, buf.value)
        if rc == 1 and c["kind"] < 2:
            assert num.value == c["number"], c["name"]
        if (rc == 1 and c["kind"] == 2) or rc == -1:
            assert buf.value == base64.b64decode(c["string"]), (c["name"], buf.value)


def test_cli_argument_and_config_errors_match_reference_cli(tmp_path):
    """Everything the CLI reports BEFORE it needs the device -- usage / help / unknown option / missing value / invalid
    enum / unreadable or malformed config / missing par file / unreadable first image (T:1518-1711) -- against the text and
    exit code of the compiled reference CLI (tests/golden/surface_cases.json)."""
    import subprocess
    g = _surface()["cli"]
    for rel, text in g["files"].items():
        os.makedirs(os.path.dirname(str(tmp_path / rel)), exist_ok=True)
        (tmp_path / rel).write_text(text)
    assert os.path.exists(pipe.CLI_PATH), "run __graft_entry__.build() first"
    for c in g["cases"]:
        p = subprocess.run([pipe.CLI_PATH] + c["args"], cwd=str(tmp_path), capture_output=True, text=True)
        assert p.returncode == c["rc"], (c["args"], p.stderr)
        assert p.stdout == c["stdout"], c["args"]
        assert p.stderr.replace(pipe.CLI_PATH, "<CLI>") == c["stderr"], c["args"]
