"""ctypes binding of include/sfmx.h (libsfmx.so) — plumbing for tests and bench.py.

There is NO fallback: if the library is missing or no gfx950 device can be opened, every entry
point raises ``SfmxError``.  Nothing here computes anything on the CPU.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

# the pipeline keeps five streams busy; HIP's default of 4 hardware queues makes lanes share one.
# Only effective if the HIP runtime has not initialised yet (bench.py sets it before importing torch).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libsfmx.so")

SFMX_OK, SFMX_ERR_INVALID, SFMX_ERR_HIP, SFMX_ERR_NO_DEVICE, SFMX_ERR_SINGULAR, SFMX_ERR_UNSUPPORTED = range(6)

# every symbol include/sfmx.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "sfmx_ctx_create", "sfmx_ctx_create_prio", "sfmx_ctx_destroy", "sfmx_last_error", "sfmx_sync", "sfmx_ctx_device", "sfmx_ctx_make_current", "sfmx_stream", "sfmx_set_timing", "sfmx_get_timing",
    "sfmx_last_kernel_us", "sfmx_kernel_profile", "sfmx_kernel_profile_name", "sfmx_pyramid_create", "sfmx_pyramid_destroy", "sfmx_pyramid_upload",
    "sfmx_pyramid_set_device", "sfmx_pyramid_set_device_async", "sfmx_pyramid_wait", "sfmx_pyramid_fetched_level", "sfmx_pyramid_download_level", "sfmx_pyramid_level_size", "sfmx_shi_tomasi_score",
    "sfmx_shi_tomasi_candidates", "sfmx_shi_tomasi_candidates_pruned", "sfmx_shi_tomasi_fetch_all_keys", "sfmx_klt_track", "sfmx_ransac_score", "sfmx_ransac_score_ex", "sfmx_sampson_mask", "sfmx_ba_create",
    "sfmx_ba_reset", "sfmx_ba_destroy", "sfmx_ba_build", "sfmx_ba_step", "sfmx_ba_begin", "sfmx_ba_end", "sfmx_ba_build_partial", "sfmx_ba_step_sharded", "sfmx_ba_step_sharded_elements", "sfmx_solve_dense", "sfmx_posegraph_solve",
    "sfmx_comm_get_unique_id", "sfmx_comm_create", "sfmx_comm_destroy", "sfmx_comm_rank", "sfmx_comm_world", "sfmx_shard_range",
    "sfmx_comm_allreduce_f64", "sfmx_comm_allreduce_u64_max",
    "sfmx_debug_hypot", "sfmx_debug_divsqrt", "sfmx_debug_klt_slow_steps",
]


class SfmxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"sfmx status {status}: {msg}")
        self.status = status


class KltCfg(ctypes.Structure):
    _fields_ = [("levels", c_int), ("win_radius", c_int), ("iters", c_int), ("fb_thresh", c_double)]


_lib = None


def load_library() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmxError(SFMX_ERR_NO_DEVICE, f"{LIB_PATH} not built: run __graft_entry__.build() (hipcc, gfx950)")
        # PyTorch-ROCm ships its own libamdhip64.so.7; load it first so that torch (device memory,
        # torch.distributed) and libsfmx share ONE HIP runtime in this process.
        if "torch" not in __import__("sys").modules and not os.environ.get("SFMX_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sfmx_last_error.restype = c_char_p
        _lib.sfmx_last_kernel_us.restype = c_double
        _lib.sfmx_stream.restype = c_void_p
        _lib.sfmx_kernel_profile_name.restype = c_char_p
        _lib.sfmx_debug_klt_slow_steps.restype = c_uint64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(POINTER(t))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Pyramid:
    def __init__(self, ctx: "Context", w: int, h: int, levels: int):
        self.ctx, self.w, self.h, self.levels = ctx, w, h, levels
        self.h_ = c_void_p()
        ctx._chk(ctx.lib.sfmx_pyramid_create(ctx.h_, c_int(w), c_int(h), c_int(levels), byref(self.h_)))

    def upload(self, img: np.ndarray):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.h, self.w)
        self.ctx._chk(self.ctx.lib.sfmx_pyramid_upload(self.ctx.h_, self.h_, _p(img, c_uint8)))
        return self

    def set_device(self, dev_ptr: int):
        self.ctx._chk(self.ctx.lib.sfmx_pyramid_set_device(self.ctx.h_, self.h_, c_void_p(dev_ptr)))
        return self

    def level(self, l: int) -> np.ndarray:
        w, h = c_int(), c_int()
        self.ctx._chk(self.ctx.lib.sfmx_pyramid_level_size(self.h_, c_int(l), byref(w), byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        self.ctx._chk(self.ctx.lib.sfmx_pyramid_download_level(self.ctx.h_, self.h_, c_int(l), _p(out, c_uint8)))
        return out

    def close(self):
        if self.h_:
            self.ctx.lib.sfmx_pyramid_destroy(self.ctx.h_, self.h_)
            self.h_ = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BaProblem:
    def __init__(self, ctx: "Context", W: int, X, obs_ptr, obs_li, obs_uv):
        self.ctx, self.W = ctx, W
        X = _f64(X)
        obs_ptr = np.ascontiguousarray(obs_ptr, np.int32)
        obs_li = np.ascontiguousarray(obs_li, np.int32)
        obs_uv = _f64(obs_uv)
        self.P = X.shape[0]
        self.h_ = c_void_p()
        ctx._chk(ctx.lib.sfmx_ba_create(ctx.h_, c_int(W), c_int(self.P), _p(X, c_double), _p(obs_ptr, c_int32),
                                        _p(obs_li, c_int32), _p(obs_uv, c_double), byref(self.h_)))

    def build(self, poses_wc, fx, fy, cx, cy, huber, lam, damp=True):
        D = 6 * self.W
        poses = _f64(poses_wc)
        S = np.zeros((D, D))
        b = np.zeros(D)
        self.ctx._chk(self.ctx.lib.sfmx_ba_build(self.ctx.h_, self.h_, _p(poses, c_double), c_double(fx), c_double(fy),
                                                 c_double(cx), c_double(cy), c_double(huber), c_double(lam),
                                                 c_int(1 if damp else 0), _p(S, c_double), _p(b, c_double)))
        return S, b

    def step(self, poses_wc, fx, fy, cx, cy, huber, lam):
        """returns (status, dx): status SFMX_OK or SFMX_ERR_SINGULAR"""
        poses = _f64(poses_wc)
        dx = np.zeros(6 * self.W)
        rc = self.ctx.lib.sfmx_ba_step(self.ctx.h_, self.h_, _p(poses, c_double), c_double(fx), c_double(fy), c_double(cx),
                                       c_double(cy), c_double(huber), c_double(lam), _p(dx, c_double))
        if rc not in (SFMX_OK, SFMX_ERR_SINGULAR):
            self.ctx._chk(rc)
        return rc, dx

    def begin(self, iters, fx, fy, cx, cy, huber, lam):
        """bracket of a job of `iters` step() calls (resident kernel for window-sized problems)"""
        self.ctx._chk(self.ctx.lib.sfmx_ba_begin(self.ctx.h_, self.h_, c_int(iters), c_double(fx), c_double(fy), c_double(cx), c_double(cy),
                                                 c_double(huber), c_double(lam)))

    def end(self):
        self.ctx._chk(self.ctx.lib.sfmx_ba_end(self.ctx.h_, self.h_))

    def step_sharded(self, comm, poses_wc, fx, fy, cx, cy, huber, lam):
        """point-sharded iteration: partial build, RCCL all-reduce of S|b in HBM, damping + gauge, solve (comm None = 1 rank)"""
        poses = _f64(poses_wc)
        dx = np.zeros(6 * self.W)
        rc = self.ctx.lib.sfmx_ba_step_sharded(self.ctx.h_, comm.h_ if comm is not None else None, self.h_, _p(poses, c_double), c_double(fx),
                                               c_double(fy), c_double(cx), c_double(cy), c_double(huber), c_double(lam), _p(dx, c_double))
        if rc not in (SFMX_OK, SFMX_ERR_SINGULAR):
            self.ctx._chk(rc)
        return rc, dx

    def step_sharded_elements(self, comm, poses_wc, fx, fy, cx, cy, huber, lam):
        """element-sharded iteration (this problem holds the WHOLE window): bit-identical to step() at any world size"""
        poses = _f64(poses_wc)
        dx = np.zeros(6 * self.W)
        rc = self.ctx.lib.sfmx_ba_step_sharded_elements(self.ctx.h_, comm.h_ if comm is not None else None, self.h_, _p(poses, c_double),
                                                        c_double(fx), c_double(fy), c_double(cx), c_double(cy), c_double(huber), c_double(lam),
                                                        _p(dx, c_double))
        if rc not in (SFMX_OK, SFMX_ERR_SINGULAR):
            self.ctx._chk(rc)
        return rc, dx

    def build_partial(self, poses_wc, fx, fy, cx, cy, huber):
        """device pointers (S, b) of this shard's raw sums — for RCCL all-reduce by the caller"""
        poses = _f64(poses_wc)
        S, b = c_void_p(), c_void_p()
        self.ctx._chk(self.ctx.lib.sfmx_ba_build_partial(self.ctx.h_, self.h_, _p(poses, c_double), c_double(fx), c_double(fy),
                                                         c_double(cx), c_double(cy), c_double(huber), byref(S), byref(b)))
        return S.value, b.value

    def close(self):
        if self.h_:
            self.ctx.lib.sfmx_ba_destroy(self.ctx.h_, self.h_)
            self.h_ = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """rank 0: the 128-byte RCCL unique id the other ranks need for Comm(...)"""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    rc = load_library().sfmx_comm_get_unique_id(buf)
    if rc != SFMX_OK:
        raise SfmxError(rc, "sfmx_comm_get_unique_id (is librccl loadable?)")
    return buf.raw


class Comm:
    """One RCCL communicator (one per host thread / lane that issues collectives)."""

    def __init__(self, device: int, unique_id: bytes | None, rank: int, world: int):
        self.lib = load_library()
        self.h_ = c_void_p()
        rc = self.lib.sfmx_comm_create(c_int(device), unique_id, c_int(rank), c_int(world), byref(self.h_))
        if rc != SFMX_OK:
            raise SfmxError(rc, "sfmx_comm_create")
        self.rank, self.world = rank, world

    def close(self):
        if self.h_:
            self.lib.sfmx_comm_destroy(self.h_)
            self.h_ = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n: int, rank: int, world: int):
    lo, hi = c_int(), c_int()
    load_library().sfmx_shard_range(c_int(n), c_int(rank), c_int(world), byref(lo), byref(hi))
    return lo.value, hi.value


class Context:
    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h_ = c_void_p()
        rc = self.lib.sfmx_ctx_create(c_int(device), byref(self.h_))
        if rc != SFMX_OK:
            raise SfmxError(rc, "sfmx_ctx_create failed: no usable MI355X (gfx950) device — there is no CPU fallback")

    def _chk(self, rc: int):
        if rc != SFMX_OK:
            raise SfmxError(rc, (self.lib.sfmx_last_error(self.h_) or b"").decode())

    def close(self):
        if self.h_:
            self.lib.sfmx_ctx_destroy(self.h_)
            self.h_ = c_void_p()

    def set_timing(self, on: bool):
        self._chk(self.lib.sfmx_set_timing(self.h_, c_int(1 if on else 0)))

    def last_kernel_us(self) -> float:
        return float(self.lib.sfmx_last_kernel_us(self.h_))

    def kernel_profile(self, reset: bool = False) -> dict:
        """{kernel name: (gpu_us, launches)} accumulated on this context since timing was enabled"""
        us = (c_double * 32)()
        calls = (c_uint64 * 32)()
        n = self.lib.sfmx_kernel_profile(self.h_, c_int(1 if reset else 0), c_int(32), us, calls)
        return {self.lib.sfmx_kernel_profile_name(c_int(i)).decode(): (float(us[i]), int(calls[i])) for i in range(n)}

    def sync(self):
        self._chk(self.lib.sfmx_sync(self.h_))

    def pyramid(self, img: np.ndarray, levels: int) -> Pyramid:
        h, w = img.shape
        return Pyramid(self, w, h, levels).upload(img)

    def shi_score(self, pyr: Pyramid):
        score = np.zeros((pyr.h, pyr.w))
        mx = c_double()
        self._chk(self.lib.sfmx_shi_tomasi_score(self.h_, pyr.h_, _p(score, c_double), byref(mx)))
        return score, mx.value

    def shi_candidates(self, pyr: Pyramid, quality: float, cap: int | None = None):
        cap = cap or pyr.w * pyr.h
        xy = np.zeros(cap, np.uint32)
        sc = np.zeros(cap)
        n = c_int()
        mx = c_double()
        self._chk(self.lib.sfmx_shi_tomasi_candidates(self.h_, pyr.h_, c_double(quality), c_int(cap), _p(xy, c_uint32),
                                                      _p(sc, c_double), byref(n), byref(mx)))
        m = min(n.value, cap)
        return (xy[:m] & 0xFFFF).astype(np.int32), (xy[:m] >> 16).astype(np.int32), sc[:m].copy(), n.value, mx.value

    def shi_candidates_pruned(self, pyr: Pyramid, quality: float, min_dist: int, cap: int | None = None):
        cap = cap or pyr.w * pyr.h
        xy = np.zeros(cap, np.uint32)
        sc = np.zeros(cap)
        n, ntot = c_int(), c_int()
        mx = c_double()
        full = np.zeros(cap, np.int32)
        self._chk(self.lib.sfmx_shi_tomasi_candidates_pruned(self.h_, pyr.h_, c_double(quality), c_int(min_dist), c_int(cap),
                                                             _p(xy, c_uint32), _p(sc, c_double), _p(full, c_int32), byref(n),
                                                             byref(ntot), byref(mx)))
        m = min(n.value, cap)
        return ((xy[:m] & 0x7FFF).astype(np.int32), ((xy[:m] >> 16) & 0x7FFF).astype(np.int32), (xy[:m] >> 31).astype(bool),
                sc[:m].copy(), n.value, ntot.value)

    def klt_track(self, pa: Pyramid, pb: Pyramid, xy, levels=3, radius=5, iters=10, fb=1.0):
        xy = _f64(xy).reshape(-1, 2)
        n = xy.shape[0]
        fwd = np.zeros((n, 2))
        back = np.zeros((n, 2))
        keep = np.zeros(n, np.uint8)
        steps = c_uint64()
        cfg = KltCfg(levels, radius, iters, fb)
        self._chk(self.lib.sfmx_klt_track(self.h_, pa.h_, pb.h_, _p(xy, c_double), c_int(n), byref(cfg), _p(fwd, c_double),
                                          _p(back, c_double), _p(keep, c_uint8), byref(steps)))
        return fwd, back, keep, steps.value

    def klt_slow_steps(self) -> int:
        return int(self.lib.sfmx_debug_klt_slow_steps(self.h_))

    def ransac_score(self, xi, xj, idx8, thr, want_E=False):
        xi, xj = _f64(xi), _f64(xj)
        idx8 = np.ascontiguousarray(idx8, np.int32)
        H = idx8.shape[0]
        counts = np.zeros(H, np.int32)
        bi, bc = c_int32(), c_int32()
        E = np.zeros((H, 3, 3)) if want_E else None
        self._chk(self.lib.sfmx_ransac_score(self.h_, _p(xi, c_double), _p(xj, c_double), c_int(xi.shape[0]), _p(idx8, c_int32),
                                             c_int(H), c_double(thr), _p(counts, c_int32), byref(bi), byref(bc),
                                             _p(E, c_double) if want_E else None))
        return counts, bi.value, bc.value, E

    def ransac_score_ex(self, xi, xj, idx8, thr):
        """dict(counts, lo, hi, flags, cond, best_iter, best_count, E): certified per-hypothesis inlier counts"""
        xi, xj = _f64(xi), _f64(xj)
        idx8 = np.ascontiguousarray(idx8, np.int32)
        H = idx8.shape[0]
        counts, lo, hi = np.zeros(H, np.int32), np.zeros(H, np.int32), np.zeros(H, np.int32)
        flags = np.zeros(H, np.uint8)
        cond = np.zeros(H)
        E = np.zeros((H, 3, 3))
        bi, bc = c_int32(), c_int32()
        self._chk(self.lib.sfmx_ransac_score_ex(self.h_, _p(xi, c_double), _p(xj, c_double), c_int(xi.shape[0]), _p(idx8, c_int32),
                                                c_int(H), c_double(thr), _p(counts, c_int32), _p(lo, c_int32), _p(hi, c_int32),
                                                _p(flags, c_uint8), _p(cond, c_double), byref(bi), byref(bc), _p(E, c_double)))
        return dict(counts=counts, lo=lo, hi=hi, flags=flags, cond=cond, best_iter=bi.value, best_count=bc.value, E=E)

    def sampson_mask(self, xi, xj, E, thr):
        xi, xj, E = _f64(xi), _f64(xj), _f64(E)
        n = xi.shape[0]
        mask = np.zeros(n, np.uint8)
        cnt = c_int32()
        self._chk(self.lib.sfmx_sampson_mask(self.h_, _p(xi, c_double), _p(xj, c_double), c_int(n), _p(E, c_double), c_double(thr),
                                             _p(mask, c_uint8), byref(cnt)))
        return mask, cnt.value

    def ba_problem(self, W, X, obs_ptr, obs_li, obs_uv) -> BaProblem:
        return BaProblem(self, W, X, obs_ptr, obs_li, obs_uv)

    def solve_dense(self, A, b):
        A, b = _f64(A), _f64(b)
        n = b.shape[0]
        x = np.zeros(n)
        rc = self.lib.sfmx_solve_dense(self.h_, _p(A, c_double), _p(b, c_double), c_int(n), _p(x, c_double))
        if rc not in (SFMX_OK, SFMX_ERR_SINGULAR):
            self._chk(rc)
        return rc, x

    def posegraph_solve(self, n, entries_ij, entries_v, g3):
        """structured pose-graph solve (tolerance mode): lower-triangle entries of L, g [n][3] -> (status, x [n][3])"""
        ij = np.ascontiguousarray(entries_ij, np.int32).reshape(-1, 2)
        v = _f64(entries_v)
        g = _f64(g3).reshape(n, 3)
        x = np.zeros((n, 3))
        rc = self.lib.sfmx_posegraph_solve(self.h_, c_int(n), _p(ij, c_int32), _p(v, c_double), c_int(len(v)), _p(g, c_double), _p(x, c_double))
        if rc not in (SFMX_OK, SFMX_ERR_SINGULAR):
            self._chk(rc)
        return rc, x

    def debug_hypot(self, x, y):
        x, y = _f64(x), _f64(y)
        out = np.zeros_like(x)
        self._chk(self.lib.sfmx_debug_hypot(self.h_, _p(x, c_double), _p(y, c_double), c_int(x.size), _p(out, c_double)))
        return out

    def debug_divsqrt(self, x, y):
        x, y = _f64(x), _f64(y)
        d, s = np.zeros_like(x), np.zeros_like(x)
        self._chk(self.lib.sfmx_debug_divsqrt(self.h_, _p(x, c_double), _p(y, c_double), c_int(x.size), _p(d, c_double),
                                              _p(s, c_double)))
        return d, s

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
