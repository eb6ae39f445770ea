"""ctypes binding of the C++ host pipeline (libsfmx_host.so: sfmx_pipeline_run) — plumbing only."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_ubyte, c_ulonglong, c_void_p

import numpy as np

from . import capi

HOST_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libsfmx_host.so")
CLI_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "templering_sfm")


class PipelineCfg(ctypes.Structure):
    _fields_ = [("frames", c_int), ("export_pointcloud", c_int), ("max_tracks", c_int), ("min_tracks", c_int),
                ("quality", c_double), ("min_distance", c_int), ("pyr_levels", c_int), ("win_radius", c_int),
                ("klt_iters", c_int), ("fb_thresh", c_double), ("kf_min_gap", c_int), ("kf_min_inliers", c_int),
                ("kf_parallax_px", c_double), ("ba_window", c_int), ("ba_iters", c_int), ("ba_max_points", c_int),
                ("ba_huber", c_double), ("ba_lambda", c_double),
                ("comm_ba", c_void_p), ("comm_ransac", c_void_p)]


class PipelineStats(ctypes.Structure):
    _fields_ = [("n_keyframes", c_int), ("n_points", c_int), ("n_edges", c_int), ("n_frames", c_int),
                ("sec_total", c_double), ("sec_klt", c_double), ("sec_shi", c_double), ("sec_ransac", c_double),
                ("sec_ba", c_double), ("sec_upload", c_double), ("sec_host", c_double),
                ("sec_shi_gpu", c_double), ("sec_shi_replay", c_double), ("sec_desc", c_double), ("sec_bookkeeping", c_double),
                ("sec_r_pre", c_double), ("sec_r_gpu", c_double), ("sec_r_verify", c_double), ("sec_r_decomp", c_double),
                ("sec_tri_iter", c_double), ("sec_tri_solve", c_double), ("sec_tri_insert", c_double),
                ("us_klt_kernel", c_double), ("us_ransac_kernel", c_double), ("us_ba_kernel", c_double),
                ("us_shi_kernel", c_double),
                ("lk_steps", c_ulonglong), ("tracks_in", c_ulonglong), ("klt_calls", c_ulonglong),
                ("ransac_calls", c_ulonglong), ("ransac_points", c_ulonglong), ("ba_calls", c_ulonglong),
                ("ba_iters", c_ulonglong), ("ransac_verified", c_ulonglong), ("shi_fallbacks", c_ulonglong),
                ("shi_calls", c_ulonglong), ("shi_memo_hits", c_ulonglong), ("shi_prefetched", c_ulonglong), ("sec_shi_wait", c_double),
                ("sec_setup", c_double), ("sec_wall", c_double), ("sec_pf_busy", c_double), ("sec_pf_gpu", c_double),
                ("sec_pf_replay", c_double), ("sec_lane_b_busy", c_double), ("sec_lane_c_busy", c_double),
                ("sec_join_wait", c_double), ("sec_ba_gather", c_double), ("sec_m_step", c_double), ("sec_m_ransac", c_double),
                ("sec_m_kf", c_double), ("sec_feed_wait", c_double), ("ransac_cert_misses", c_ulonglong),
                ("us_kernel", c_double * 16), ("calls_kernel", c_ulonglong * 16), ("sec_lane_a_busy", c_double), ("sec_lane_e_busy", c_double)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("us_kernel", "calls_kernel")}
        lib = capi.load_library()
        # per-kernel GPU time (us) and launches over every context of the run; filled only when timing was enabled
        d["kernels"] = {lib.sfmx_kernel_profile_name(c_int(i)).decode(): (float(self.us_kernel[i]), int(self.calls_kernel[i]))
                        for i in range(16) if lib.sfmx_kernel_profile_name(c_int(i))}
        return d


DEFAULTS = dict(frames=12, export_pointcloud=1, max_tracks=2200, min_tracks=900, quality=0.01, min_distance=8,
                pyr_levels=3, win_radius=5, klt_iters=10, fb_thresh=1.0, kf_min_gap=1, kf_min_inliers=200,
                kf_parallax_px=18.0, ba_window=6, ba_iters=5, ba_max_points=600, ba_huber=3.0, ba_lambda=1e-3)

_host = None


def load_host_library() -> ctypes.CDLL:
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise capi.SfmxError(capi.SFMX_ERR_NO_DEVICE, f"{HOST_LIB_PATH} not built: run __graft_entry__.build()")
        capi.load_library()
        _host = ctypes.CDLL(HOST_LIB_PATH)
        _host.sfmx_host_hypot.restype = c_double
        _host.sfmx_host_hypot.argtypes = [c_double, c_double]
    return _host


def run(ctx: capi.Context, images: np.ndarray | None, names, K, lat=None, lon=None, cfg: dict | None = None,
        out_dir: str | None = None, images_dev: int | None = None, shape=None, timing: bool = False, comms=None):
    """Run the per-frame loop.  images: host [F,h,w] u8, or images_dev: device pointer with shape=(F,h,w).
    comms (optional): (ba, ransac) capi.Comm objects -- every rank runs the same sequence, BA points and RANSAC hypotheses
    are sharded over the ranks: `ba` carries the S | b all-reduce of lane B, `ransac` the winner merges the geometry
    thread issues in program order (csrc/host/pipeline.hpp: PipelineConfig)."""
    lib = load_host_library()
    if images is not None:
        images = np.ascontiguousarray(images, np.uint8)
        F, h, w = images.shape
    else:
        F, h, w = shape
    c = PipelineCfg(**{**DEFAULTS, **(cfg or {})})
    if comms is not None:
        if len(comms) != 2:
            raise ValueError("comms = (ba, ransac)")
        c.comm_ba, c.comm_ransac = [m.h_ if m is not None else None for m in comms]
    arr = (c_char_p * F)(*[str(n).encode() for n in names])
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    lat = np.zeros(F) if lat is None else np.ascontiguousarray(lat, np.float64)
    lon = np.zeros(F) if lon is None else np.ascontiguousarray(lon, np.float64)
    has_ang = np.ones(F, np.uint8)
    log = ctypes.create_string_buffer(1 << 20)
    st = PipelineStats()
    centres = np.zeros((F, 3))
    ctx.set_timing(timing)
    rc = lib.sfmx_pipeline_run(ctx.h_, images.ctypes.data_as(c_void_p) if images is not None else None,
                               c_void_p(images_dev) if images_dev else None, c_int(F), c_int(w), c_int(h), arr,
                               K.ctypes.data_as(POINTER(c_double)), lat.ctypes.data_as(POINTER(c_double)),
                               lon.ctypes.data_as(POINTER(c_double)), has_ang.ctypes.data_as(POINTER(c_ubyte)), byref(c),
                               out_dir.encode() if out_dir else None, log, c_int(len(log)), byref(st),
                               centres.ctypes.data_as(POINTER(c_double)), c_int(F))
    text = log.value.decode()
    if rc != capi.SFMX_OK:
        raise capi.SfmxError(rc, text.strip())
    return dict(log=text, stats=st.asdict(), centres=centres[:st.n_keyframes].copy())


def find_E_ransac(ctx: capi.Context, K, pi, pj, iters: int, thr: float, min_inliers: int):
    """The find_E_ransac seam (T:646-761) of the host library on its own: dict(ok, R, t, inliers, best_iter)."""
    lib = load_host_library()
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    pi = np.ascontiguousarray(pi, np.float64)
    pj = np.ascontiguousarray(pj, np.float64)
    n = pi.shape[0]
    R, t = np.zeros((3, 3)), np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    n_inl, best = c_int(0), c_int(-1)
    dp = POINTER(c_double)
    rc = lib.sfmx_host_find_E_ransac(ctx.h_, K.ctypes.data_as(dp), pi.ctypes.data_as(dp), pj.ctypes.data_as(dp), c_int(n), c_int(iters),
                                     c_double(thr), c_int(min_inliers), R.ctypes.data_as(dp), t.ctypes.data_as(dp),
                                     inl.ctypes.data_as(POINTER(c_int)), byref(n_inl), byref(best))
    if rc < 0:
        raise capi.SfmxError(-rc, "find_E_ransac")
    return dict(ok=rc, R=R, t=t, inliers=inl[:n_inl.value].copy(), best_iter=best.value)


def find_E_ransac_world(ctx: capi.Context, K, pi, pj, iters: int, thr: float, min_inliers: int, world: int, as_rank: int):
    """find_E_ransac as `world` ranks run it, emulated on one GPU (test hook): each virtual rank's local winner, the merge the
    all-reduces compute, and the result as rank `as_rank` forms it."""
    lib = load_host_library()
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    pi = np.ascontiguousarray(pi, np.float64)
    pj = np.ascontiguousarray(pj, np.float64)
    n = pi.shape[0]
    R, t = np.zeros((3, 3)), np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    n_inl, best = c_int(0), c_int(-1)
    dp = POINTER(c_double)
    rc = lib.sfmx_host_find_E_ransac_world(ctx.h_, K.ctypes.data_as(dp), pi.ctypes.data_as(dp), pj.ctypes.data_as(dp), c_int(n), c_int(iters),
                                           c_double(thr), c_int(min_inliers), c_int(world), c_int(as_rank), R.ctypes.data_as(dp),
                                           t.ctypes.data_as(dp), inl.ctypes.data_as(POINTER(c_int)), byref(n_inl), byref(best))
    if rc < 0:
        raise capi.SfmxError(-rc, "find_E_ransac_world")
    return dict(ok=rc, R=R, t=t, inliers=inl[:n_inl.value].copy(), best_iter=best.value)


class Tracker:
    """The KLTTracker seam (T:307-400) of the host library on its own: step(image) -> (prev, cur, ids); tracks()."""

    def __init__(self, ctx: capi.Context, w: int, h: int, max_tracks=2200, min_tracks=900, quality=0.01, min_distance=8, levels=3,
                 radius=5, iters=10, fb=1.0):
        self.lib = load_host_library()
        self.cap, self.w, self.h = max_tracks, w, h
        self.lib.sfmx_host_tracker_create.restype = c_void_p
        self.h_ = c_void_p(self.lib.sfmx_host_tracker_create(ctx.h_, c_int(w), c_int(h), c_int(max_tracks), c_int(min_tracks),
                                                              c_double(quality), c_int(min_distance), c_int(levels), c_int(radius),
                                                              c_int(iters), c_double(fb)))
        if not self.h_:
            raise capi.SfmxError(capi.SFMX_ERR_INVALID, "tracker_create")

    def step(self, img: np.ndarray):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.h, self.w)
        prev, cur = np.zeros((self.cap, 2)), np.zeros((self.cap, 2))
        ids = np.zeros(self.cap, np.int32)
        dp = POINTER(c_double)
        n = self.lib.sfmx_host_tracker_step(self.h_, img.ctypes.data_as(c_void_p), prev.ctypes.data_as(dp), cur.ctypes.data_as(dp),
                                            ids.ctypes.data_as(POINTER(c_int)), c_int(self.cap))
        if n < 0:
            raise capi.SfmxError(-n, "tracker_step")
        return prev[:n].copy(), cur[:n].copy(), ids[:n].copy()

    def tracks(self):
        xy = np.zeros((self.cap, 2))
        ids = np.zeros(self.cap, np.int32)
        n = self.lib.sfmx_host_tracker_tracks(self.h_, xy.ctypes.data_as(POINTER(c_double)), ids.ctypes.data_as(POINTER(c_int)), c_int(self.cap))
        if n < 0:
            raise capi.SfmxError(-n, "tracker_tracks")
        return xy[:n].copy(), ids[:n].copy()

    def close(self):
        if self.h_:
            self.lib.sfmx_host_tracker_destroy(self.h_)
            self.h_ = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def posegraph(ctx: capi.Context, Rs, centres, ei, ej, eR, et, is_loop):
    """posegraph_optimize_centers (T:1131-1197) of the host library on its own: (ok, centres)"""
    lib = load_host_library()
    Rs = np.ascontiguousarray(Rs, np.float64)
    c = np.ascontiguousarray(centres, np.float64).copy()
    ei, ej = np.ascontiguousarray(ei, np.int32), np.ascontiguousarray(ej, np.int32)
    eR, et = np.ascontiguousarray(eR, np.float64), np.ascontiguousarray(et, np.float64)
    lp = np.ascontiguousarray(is_loop, np.int32)
    dp, ip = POINTER(c_double), POINTER(c_int)
    rc = lib.sfmx_host_posegraph(ctx.h_, c_int(len(c)), Rs.ctypes.data_as(dp), c.ctypes.data_as(dp), c_int(len(ei)), ei.ctypes.data_as(ip),
                                 ej.ctypes.data_as(ip), eR.ctypes.data_as(dp), et.ctypes.data_as(dp), lp.ctypes.data_as(ip))
    if rc < 0:
        raise capi.SfmxError(-rc, "posegraph_optimize_centers")
    return rc, c
