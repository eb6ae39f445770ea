"""Test-side loaders for the checker libraries (TEST INFRASTRUCTURE).

* ``oracle()``  -> oracle/_build/libsfm_oracle.so  (CPU restatement, travels with the repo)
* ``ref()``     -> oracle/_ref/libsfmref.so        (the real reference, built only where
                   /root/reference exists; ``None`` when absent)

Both are called through a tiny auto-marshalling shim: numpy arrays go as pointers,
Python ints as ``c_int``, floats as ``c_double``.
"""
from __future__ import annotations

import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "structure-from-motion-3d-reconstruction_amd"

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pkg():
    return importlib.import_module(PKG_NAME)


class CLib:
    """ctypes library with numpy-aware calls: lib.call("fn", restype, *args)."""

    def __init__(self, path: str):
        self.path = path
        self.dll = ctypes.CDLL(path)

    def has(self, name: str) -> bool:
        return hasattr(self.dll, name)

    def call(self, name: str, restype, *args):
        fn = getattr(self.dll, name)
        fn.restype = restype
        conv = []
        keep = []
        for a in args:
            if isinstance(a, np.ndarray):
                assert a.flags["C_CONTIGUOUS"], f"{name}: non-contiguous array"
                conv.append(a.ctypes.data_as(ctypes.c_void_p))
                keep.append(a)
            elif a is None:
                conv.append(ctypes.c_void_p(None))
            elif isinstance(a, (bool, np.bool_)):
                conv.append(ctypes.c_int(int(a)))
            elif isinstance(a, (int, np.integer)):
                conv.append(ctypes.c_int(int(a)))
            elif isinstance(a, (float, np.floating)):
                conv.append(ctypes.c_double(float(a)))
            elif isinstance(a, bytes):
                conv.append(ctypes.c_char_p(a))
            else:
                conv.append(a)
        return fn(*conv)


_cache: dict[str, CLib | None] = {}


def _build_oracle() -> str:
    so = os.path.join(ROOT, "oracle", "_build", "libsfm_oracle.so")
    src = os.path.join(ROOT, "oracle", "sfm_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"],
                              stdout=subprocess.DEVNULL)
    return so


def oracle() -> CLib:
    if "oracle" not in _cache:
        _cache["oracle"] = CLib(_build_oracle())
    return _cache["oracle"]


def ref() -> CLib | None:
    if "ref" not in _cache:
        so = os.path.join(ROOT, "oracle", "_ref", "libsfmref.so")
        _cache["ref"] = CLib(so) if os.path.exists(so) else None
    return _cache["ref"]


def ref_cli() -> str | None:
    p = os.path.join(ROOT, "oracle", "_ref", "templering_sfm_ref")
    return p if os.path.exists(p) else None


# ------------------------------------------------------------------ convenience wrappers
# Each takes lib (oracle() or ref()) and a function-name prefix ("orc" / "ref").

def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def bits(a) -> np.ndarray:
    """View float64 data as uint64 so comparisons are bit-exact (NaN == NaN, -0 != +0)."""
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits_equal(a, b, what="", nan_equal=False):
    """nan_equal: any NaN matches any NaN (x86 generates the negative 'indefinite' QNaN, gfx950 the
    positive canonical one; NaN-ness is part of the parity contract, its sign/payload is not)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ne = a.view(np.uint64) != b.view(np.uint64)
    if nan_equal:
        ne &= ~(np.isnan(a) & np.isnan(b))
    if ne.any():
        idx = np.argwhere(ne)[:5]
        raise AssertionError(
            f"{what}: {int(ne.sum())}/{ne.size} FP64 values differ bitwise; first at {idx.tolist()} "
            f"a={a[tuple(idx[0])]!r} b={b[tuple(idx[0])]!r}")


def downsample2(lib, pre, img):
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), np.uint8)
    lib.call(f"{pre}_downsample2", None, u8(img), w, h, out)
    return out


def shi_tomasi(lib, pre, img, max_corners, quality, min_dist):
    h, w = img.shape
    out = np.zeros((max(max_corners, 1), 2), np.float64)
    n = lib.call(f"{pre}_shi_tomasi", ctypes.c_int, u8(img), w, h, max_corners, float(quality), min_dist, out)
    return out[:n].copy()


def lk_step(lib, pre, i0, i1, radius, x, y):
    h, w = i0.shape
    out = np.zeros(2)
    lib.call(f"{pre}_lk_step", None, u8(i0), u8(i1), w, h, radius, float(x), float(y), out)
    return out


def klt_track(lib, pre, ia, ib, levels, radius, iters, xy, fb=1.0):
    h, w = ia.shape
    xy = f64(xy)
    n = xy.shape[0]
    fwd = np.zeros((n, 2))
    back = np.zeros((n, 2))
    if pre == "orc":
        keep = np.zeros(n, np.uint8)
        lib.call("orc_klt_track", None, u8(ia), u8(ib), w, h, levels, radius, iters, float(fb), n, xy, fwd, back, keep)
        return fwd, back, keep
    lib.call("ref_klt_track", None, u8(ia), u8(ib), w, h, levels, radius, iters, n, xy, fwd, back)
    keep = (~(np.hypot(back[:, 0] - xy[:, 0], back[:, 1] - xy[:, 1]) >= fb)).astype(np.uint8)
    return fwd, back, keep


class Tracker:
    def __init__(self, lib, pre, max_tracks=2200, min_tracks=900, quality=0.01, min_distance=8, levels=3,
                 radius=5, iters=10, fb=1.0):
        self.lib, self.pre, self.cap = lib, pre, max_tracks
        fn = getattr(lib.dll, f"{pre}_tracker_create")
        fn.restype = ctypes.c_void_p
        self.h = ctypes.c_void_p(fn(ctypes.c_int(max_tracks), ctypes.c_int(min_tracks), ctypes.c_double(quality),
                                    ctypes.c_int(min_distance), ctypes.c_int(levels), ctypes.c_int(radius),
                                    ctypes.c_int(iters), ctypes.c_double(fb)))

    def step(self, img):
        h, w = img.shape
        prev = np.zeros((self.cap, 2))
        cur = np.zeros((self.cap, 2))
        ids = np.zeros(self.cap, np.int32)
        n = self.lib.call(f"{self.pre}_tracker_step", ctypes.c_int, self.h, u8(img), w, h, prev, cur, ids)
        return prev[:n].copy(), cur[:n].copy(), ids[:n].copy()

    def tracks(self):
        xy = np.zeros((self.cap, 2))
        ids = np.zeros(self.cap, np.int32)
        n = self.lib.call(f"{self.pre}_tracker_tracks", ctypes.c_int, self.h, xy, ids)
        return xy[:n].copy(), ids[:n].copy()

    def close(self):
        if self.h:
            self.lib.call(f"{self.pre}_tracker_destroy", None, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def uniform_draws(lib, pre, seed, n, count):
    out = np.zeros(count, np.int32)
    lib.call(f"{pre}_uniform_draws", None, ctypes.c_uint(seed), n, count, out)
    return out


def normalize_points(lib, pre, K, px):
    px = f64(px)
    out = np.zeros_like(px)
    rc = lib.call(f"{pre}_normalize_points", ctypes.c_int, f64(K).reshape(9), px, px.shape[0], out)
    return rc, out


def jacobi(lib, pre, A, iters):
    A = f64(A)
    n = A.shape[0]
    w = np.zeros(n)
    V = np.zeros((n, n))
    lib.call(f"{pre}_jacobi_eig_sym", None, A, n, iters, w, V)
    return w, V


def svd3(lib, pre, A):
    U, s, V = np.zeros((3, 3)), np.zeros(3), np.zeros((3, 3))
    lib.call(f"{pre}_svd3", None, f64(A), U, s, V)
    return U, s, V


def eight_point(lib, pre, xn, yn, idx8):
    E = np.zeros((3, 3))
    lib.call(f"{pre}_eight_point_E", None, f64(xn), f64(yn), len(xn), i32(idx8), E)
    return E


def sampson(lib, pre, E, x, xp):
    return lib.call(f"{pre}_sampson_err", ctypes.c_double, f64(E), float(x[0]), float(x[1]), float(xp[0]), float(xp[1]))


def find_E_ransac(lib, pre, K, pi, pj, iters, thr, min_inliers):
    pi, pj = f64(pi), f64(pj)
    n = pi.shape[0]
    R, t = np.zeros((3, 3)), np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    n_inl = np.zeros(1, np.int32)
    if pre == "orc":
        diag = np.zeros(3, np.int32)
        E = np.zeros((3, 3))
        ok = lib.call("orc_find_E_ransac", ctypes.c_int, f64(K).reshape(9), pi, pj, n, iters, float(thr), min_inliers,
                      R, t, inl, n_inl, diag, E)
        return dict(ok=ok, R=R, t=t, inliers=inl[:n_inl[0]].copy(), best_iter=int(diag[0]), cand=int(diag[1]), E=E)
    ok = lib.call("ref_find_E_ransac", ctypes.c_int, f64(K).reshape(9), pi, pj, n, iters, float(thr), min_inliers,
                  R, t, inl, n_inl)
    return dict(ok=ok, R=R, t=t, inliers=inl[:n_inl[0]].copy())


def triangulate(lib, pre, K, Ri, ti, Rj, tj, ui, uj):
    X = np.zeros(3)
    lib.call(f"{pre}_triangulate_dlt", None, f64(K).reshape(9), f64(Ri), f64(ti), f64(Rj), f64(tj), f64(ui), f64(uj), X)
    return X


def solve_gauss(lib, pre, A, b):
    A, b = f64(A), f64(b)
    n = b.shape[0]
    x = np.zeros(n)
    rc = lib.call(f"{pre}_solve_gauss", ctypes.c_int, A, b, n, x)
    return rc, x


def so3_exp(lib, pre, w):
    R = np.zeros((3, 3))
    lib.call(f"{pre}_so3_exp", None, f64(w), R)
    return R


def so3_log(lib, pre, R):
    w = np.zeros(3)
    lib.call(f"{pre}_so3_log", None, f64(R), w)
    return w


def bundle_adjust_window(lib, pre, K, poses12, X, obs_ptr, obs_kf, obs_uv, window, iters, max_points, huber, lam):
    poses = f64(poses12).copy()
    lib.call(f"{pre}_bundle_adjust_window", None, f64(K).reshape(9), poses.shape[0], poses, len(X), f64(X),
             i32(obs_ptr), i32(obs_kf), f64(obs_uv), window, iters, max_points, float(huber), float(lam))
    return poses


def map_iteration_order(lib, pre, n):
    out = np.zeros(n, np.int32)
    lib.call(f"{pre}_map_iteration_order", None, n, out)
    return out


def posegraph(lib, pre, Rs, centres, ei, ej, eR, et, is_loop):
    c = f64(centres).copy()
    ok = lib.call(f"{pre}_posegraph_optimize_centers", ctypes.c_int, len(c), f64(Rs), c, len(ei), i32(ei), i32(ej),
                  f64(eR), f64(et), i32(is_loop))
    return ok, c


def global_desc(lib, pre, img):
    h, w = img.shape
    out = np.zeros(1024, np.float32)
    lib.call(f"{pre}_global_desc_32", None, u8(img), w, h, out)
    return out


# ------------------------------------------------------------------ whole-pipeline runner (oracle)
class OrcPipelineCfg(ctypes.Structure):
    _fields_ = [("frames", ctypes.c_int), ("export_pointcloud", ctypes.c_int),
                ("max_tracks", ctypes.c_int), ("min_tracks", ctypes.c_int), ("quality", ctypes.c_double),
                ("min_distance", ctypes.c_int), ("pyr_levels", ctypes.c_int), ("win_radius", ctypes.c_int),
                ("klt_iters", ctypes.c_int), ("fb_thresh", ctypes.c_double),
                ("kf_min_gap", ctypes.c_int), ("kf_min_inliers", ctypes.c_int), ("kf_parallax_px", ctypes.c_double),
                ("ba_window", ctypes.c_int), ("ba_iters", ctypes.c_int), ("ba_max_points", ctypes.c_int),
                ("ba_huber", ctypes.c_double), ("ba_lambda", ctypes.c_double)]


PIPE_DEFAULTS = dict(frames=12, export_pointcloud=1, max_tracks=2200, min_tracks=900, quality=0.01, min_distance=8,
                     pyr_levels=3, win_radius=5, klt_iters=10, fb_thresh=1.0, kf_min_gap=1, kf_min_inliers=200,
                     kf_parallax_px=18.0, ba_window=6, ba_iters=5, ba_max_points=600, ba_huber=3.0, ba_lambda=1e-3)


def pipe_cfg_from_json(cfg_json: dict, frames=None) -> dict:
    """Apply a reference-style config.json (cpp.* overrides common.*, T:1631-1676) to the defaults."""
    d = dict(PIPE_DEFAULTS)

    def pick(*path):
        for sec in ("cpp", "common"):
            cur = cfg_json.get(sec, {})
            ok = True
            for k in path:
                if isinstance(cur, dict) and k in cur:
                    cur = cur[k]
                else:
                    ok = False
                    break
            if ok:
                return cur
        return None
    m = {("system", "frames"): "frames", ("klt", "max_tracks"): "max_tracks", ("klt", "min_tracks"): "min_tracks",
         ("klt", "quality"): "quality", ("klt", "min_distance"): "min_distance", ("klt", "pyr_levels"): "pyr_levels",
         ("klt", "win_radius"): "win_radius", ("klt", "iters"): "klt_iters", ("klt", "fb_thresh"): "fb_thresh",
         ("keyframe", "min_gap"): "kf_min_gap", ("keyframe", "min_inliers"): "kf_min_inliers",
         ("keyframe", "parallax_px"): "kf_parallax_px", ("ba", "window"): "ba_window", ("ba", "iters"): "ba_iters",
         ("ba", "max_points"): "ba_max_points", ("ba", "huber_delta"): "ba_huber", ("ba", "lambda"): "ba_lambda"}
    for path, key in m.items():
        v = pick(*path)
        if v is not None:
            d[key] = type(PIPE_DEFAULTS[key])(v)
    if frames is not None:
        d["frames"] = frames
    return d


def orc_pipeline_run(images, names, K, lat, lon, cfg: dict, out_dir: str):
    lib = oracle()
    images = u8(images)
    F, h, w = images.shape
    c = OrcPipelineCfg(**cfg)
    arr = (ctypes.c_char_p * F)(*[n.encode() for n in names])
    log = ctypes.create_string_buffer(1 << 20)
    nk, npnt = ctypes.c_int(0), ctypes.c_int(0)
    has_ang = np.ones(F, np.uint8)
    fn = lib.dll.orc_pipeline_run
    fn.restype = ctypes.c_int
    rc = fn(images.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(F), ctypes.c_int(w), ctypes.c_int(h), arr,
            f64(K).ctypes.data_as(ctypes.c_void_p), f64(lat).ctypes.data_as(ctypes.c_void_p),
            f64(lon).ctypes.data_as(ctypes.c_void_p), has_ang.ctypes.data_as(ctypes.c_void_p), ctypes.byref(c),
            out_dir.encode(), log, ctypes.c_int(len(log)), ctypes.byref(nk), ctypes.byref(npnt), None)
    return rc, log.value.decode(), nk.value, npnt.value
