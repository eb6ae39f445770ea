// ba.hip — local bundle adjustment: reduced camera system build (T:893-1071) and dense solve
// (cpp/include/dense.hpp:54-93) for gfx950.
//
// The reference accumulates S (6W x 6W) and b point after point, so every element of S,b is an
// ORDERED floating-point sum over the points; parity is bit-exact, so the order is kept:
//
//  k_ba_points_window / k_ba_points_bulk: residuals, analytic Jacobians, Huber weights, per-point Hpp/bp and
//      per-(point,pose) Hxx/bx/Hxp exactly as T:920-1009, then inv3(Hpp) and G = Hxp * Hpp^-1, G*bp (T:1011-1041).
//      Output is one compact RECORD per point (84 doubles per observing pose) plus a pose->slot table; poses (W x 12
//      doubles) sit in LDS.  Window shape: sixteen lanes per point (one per observation), followed in the same launch by
//      (k_ba_points_window_lds: the same with observation lists, slot tables and slot records in LDS -- the window default)
//  the expansion (k_ba_expand for large problems): every addend the reference will add into S,b for a point -- the
//      Schur term G_a * Hxp_b^T per element, Hxx, bx, G*bp -- written as one contiguous contribution row per point.
//  k_ba_reduce  (16 elements of S | b per workgroup): ordered column sums over the contribution rows in the reference's
//      sequence (T:1015-1057): Hxx first, then the Schur term (which the reference ADDS, quirk Q6), bx then -G*bp for
//      b; finally damping and gauge (T:1064-1071).  Tiles of rows stream through a register ring and LDS ahead of the
//      dependent add chains, which are the only serial part.  For windows of 6 / 10 poses the workgroup that finishes
//      last also solves the system (solve_regs_wave) and publishes dx to pinned host memory.
//  k_solve_regs / k_solve_wave (one wavefront) and the blocked k_lu_* kernels (whole device): partial-pivoting
//      elimination with the reference's first-maximum pivot rule, row normalisation, |f| < 1e-18 skip and ascending
//      back-substitution.
//
// Algorithmic bytes per BA iteration: 20*R + 24*P + 96*W read, 8*(D^2+D) written.
#include "sfmx_internal.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

// ---- diagnostic build only (make HIPFLAGS_EXTRA=-DSFMX_BA_WGSTAMPS): every workgroup of the two window kernels appends
// {start, end (s_memrealtime, 100 MHz), hardware id, kind | block | grid} to a device ring, sfmx_debug_dump_wgstamps(path) writes
// it out -- where the 3x stretch of these kernels inside the pipeline comes from (tools/ba_wgstamps.py).  No code in the product build.
#ifdef SFMX_BA_WGSTAMPS
#define WGS_CAP (1u << 20)
__device__ unsigned long long g_wgs_ring[WGS_CAP][8];
__device__ unsigned g_wgs_next;
struct WgStamp {
  unsigned long long t0, m[4] = {0, 0, 0, 0};
  unsigned hw;
  __device__ __forceinline__ void mark(int i) { m[i] = __builtin_amdgcn_s_memrealtime(); }
  __device__ __forceinline__ WgStamp() {
    t0 = __builtin_amdgcn_s_memrealtime();
    hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 28);  // HW_ID | XCC_ID << 28
  }
  __device__ __forceinline__ void done(int kind, int blk, int nblk) const {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned slot = atomicAdd(&g_wgs_next, 1u) & (WGS_CAP - 1);
      g_wgs_ring[slot][0] = t0;
      g_wgs_ring[slot][1] = __builtin_amdgcn_s_memrealtime();
      g_wgs_ring[slot][2] = hw;
      g_wgs_ring[slot][3] = ((unsigned long long)kind << 48) | ((unsigned long long)(unsigned)blk << 24) | (unsigned)nblk;
      for (int i = 0; i < 4; i++) g_wgs_ring[slot][4 + i] = m[i];
    }
  }
};
extern "C" int sfmx_debug_dump_wgstamps(const char* path) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wgs_next), 4) != hipSuccess) return -1;
  const unsigned cnt = n < WGS_CAP ? n : WGS_CAP;
  std::vector<unsigned long long> h((size_t)cnt * 8);
  if (cnt && hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_wgs_ring), (size_t)cnt * 64) != hipSuccess) return -1;
  FILE* f = fopen(path, "wb");
  if (!f) return -1;
  fwrite(h.data(), 64, cnt, f);
  fclose(f);
  n = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wgs_next), &n, 4);
  return (int)cnt;
}
#define WGS_BEGIN WgStamp wgs_
#define WGS_END(kind, blk, nblk) wgs_.done(kind, blk, nblk)
#define WGS_MARK(i) wgs_.mark(i)
#define WGS_PARAM , WgStamp& wgs_
#define WGS_ARG , wgs_
#else
#define WGS_BEGIN
#define WGS_END(kind, blk, nblk)
#define WGS_MARK(i)
#define WGS_PARAM
#define WGS_ARG
#endif

#define BA_SLOT 84  // doubles per (point, pose) slot: Hxx 36 | bx 6 | Hxp 18 | G 18 | G*bp 6
#define BA_MAX_OBS 16
#define BA_MAX_W 64

struct sfmx_ba_problem {
  int W = 0, P = 0, R = 0, MS = 0;
  DevBuf bufs[12];  // grow-only backing stores, so a problem object can be reset for every BA call ([11]: shard partials of the
                    // virtual-world test mode, allocated on first use)
  double* X = nullptr;
  int32_t* obs_ptr = nullptr;
  int32_t* obs_li = nullptr;
  double* obs_uv = nullptr;
  double* poses = nullptr;    // [W][12]
  double* rec = nullptr;      // [P][MS][BA_SLOT]
  int8_t* slot_of = nullptr;  // [P][W]  (-1: pose does not see the point / point skipped)
  double* S = nullptr;        // [D*D]
  double* b = nullptr;        // [D]
  double* work = nullptr;     // solve scratch: dx [D] + status
  double* contrib = nullptr;  // [P][CS] per-point contribution rows
  unsigned* ticket = nullptr;  // finished-workgroup counter of the fused reduce + solve (zero between launches)
  hipEvent_t ev_sync = nullptr, ev_exp[2] = {nullptr, nullptr}, ev_red[2] = {nullptr, nullptr};  // chunked expand / reduce pipeline
  int chunk = 0;                // points per chunk of the contribution-row ring (0: all rows resident)
  bool lds_points = false;      // window shape without a point that one pose observes twice: k_ba_points_window_lds
  // resident job (k_ba_window_resident): launched by sfmx_ba_begin, fed by sfmx_ba_step, released by sfmx_ba_end
  bool job_active = false;
  int job_iters = 0, job_done = 0;
  unsigned long long job_first_seq = 0;
  double job_k[6] = {0, 0, 0, 0, 0, 0};  // fx, fy, cx, cy, huber, lambda the kernel was launched with
  DevBuf job_ctl;                        // BaResidentCtl
  PinBuf job_cmd;                        // command words (pinned host memory the kernel polls): a ring, one word per job, so that a
  unsigned job_slot = 0;                 // kernel still leaving on EXIT never sees the next job's commands
};

// dense.hpp:96-119
__device__ __forceinline__ bool inv3_ref(const double* A, double* inv) {
  const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
  const double c11 = (e * i - f * h), c12 = -(d * i - f * g), c13 = (d * h - e * g);
  const double c21 = -(b * i - c * h), c22 = (a * i - c * g), c23 = -(a * h - b * g);
  const double c31 = (b * f - c * e), c32 = -(a * f - c * d), c33 = (a * e - b * d);
  const double det = a * c11 + b * c12 + c * c13;
  if (fabs(det) < 1e-15) return false;
  const double r = 1.0 / det;
  inv[0] = c11 * r; inv[1] = c21 * r; inv[2] = c31 * r;
  inv[3] = c12 * r; inv[4] = c22 * r; inv[5] = c32 * r;
  inv[6] = c13 * r; inv[7] = c23 * r; inv[8] = c33 * r;
  return true;
}

// Hxx (6x6), bx (6) and Hxp (6x3) of one residual added into a slot record (T:990-1009).  FRESH: the slot was created
// by this very observation, so the reference's "zero-initialised block += v" is evaluated as 0.0 + v and stored without
// reading the record back; otherwise (the same pose observes the point twice) it is a read-modify-write.
template <bool FRESH>
__device__ __forceinline__ void ba_accumulate_slot(double* __restrict__ A, const double* Jx, const double* Jp, double wgt, double rx, double ry) {
#pragma unroll
  for (int a = 0; a < 6; a++) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
      double s = 0.0;
      s += Jx[a] * Jx[c];
      s += Jx[6 + a] * Jx[6 + c];
      A[a * 6 + c] = (FRESH ? 0.0 : A[a * 6 + c]) + wgt * s;
    }
    double sb = 0.0;
    sb += Jx[a] * rx;
    sb += Jx[6 + a] * ry;
    A[36 + a] = (FRESH ? 0.0 : A[36 + a]) + wgt * sb;
  }
#pragma unroll
  for (int a = 0; a < 6; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      s += Jx[a] * Jp[c];
      s += Jx[6 + a] * Jp[3 + c];
      A[42 + a * 3 + c] = (FRESH ? 0.0 : A[42 + a * 3 + c]) + wgt * s;
    }
}

// One observation of a point (T:921-1009): the slot record A (Hxx | bx | Hxp) and the twelve terms wgt*s it adds to Hpp (9)
// and bp (3).  false: the point is on or behind the camera plane (T:933; NaN passes, as in the reference) -- the slot
// exists (T:925-930) but stays zero and nothing is added to Hpp / bp.
template <bool FRESH>
__device__ __forceinline__ bool ba_observation(double* __restrict__ A, const double* __restrict__ R, double Xx, double Xy, double Xz, double u,
                                               double v, double fx, double fy, double cx, double cy, double huber, double* __restrict__ term) {
  const double Xcx = (R[0] * Xx + R[1] * Xy + R[2] * Xz) + R[9];
  const double Xcy = (R[3] * Xx + R[4] * Xy + R[5] * Xz) + R[10];
  const double Xcz = (R[6] * Xx + R[7] * Xy + R[8] * Xz) + R[11];
  if (Xcz <= 1e-6) {
    if (FRESH)
      for (int k = 0; k < 60; k++) A[k] = 0.0;
    return false;
  }
  const double qx = Xcx / Xcz, qy = Xcy / Xcz;
  const double rx = u - (fx * qx + cx);
  const double ry = v - (fy * qy + cy);
  const double rn = sfmx::hypot_glibc(rx, ry);
  const double wgt = (rn <= huber) ? 1.0 : huber / (rn + 1e-12);
  const double iz = 1.0 / Xcz, iz2 = iz * iz;
  double Jq[6];
  Jq[0] = fx * iz; Jq[1] = 0.0; Jq[2] = -fx * Xcx * iz2;
  Jq[3] = 0.0; Jq[4] = fy * iz; Jq[5] = -fy * Xcy * iz2;
  double Jp[6], Jr[6];
#pragma unroll
  for (int row = 0; row < 2; ++row)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double a0 = Jq[row * 3 + 0] * R[c], a1 = Jq[row * 3 + 1] * R[3 + c], a2 = Jq[row * 3 + 2] * R[6 + c];
      Jp[row * 3 + c] = a0 + a1 + a2;
    }
  const double Xm[9] = {0.0, -Xcz, Xcy, Xcz, 0.0, -Xcx, -Xcy, Xcx, 0.0};
#pragma unroll
  for (int row = 0; row < 2; ++row)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double a0 = -Jq[row * 3 + 0] * Xm[c], a1 = -Jq[row * 3 + 1] * Xm[3 + c], a2 = -Jq[row * 3 + 2] * Xm[6 + c];
      Jr[row * 3 + c] = a0 + a1 + a2;
    }
  const double Jx[12] = {Jr[0], Jr[1], Jr[2], Jq[0], Jq[1], Jq[2], Jr[3], Jr[4], Jr[5], Jq[3], Jq[4], Jq[5]};
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      double s = 0.0;
      s += Jp[a] * Jp[c];
      s += Jp[3 + a] * Jp[3 + c];
      term[a * 3 + c] = wgt * s;
    }
    double sb = 0.0;
    sb += Jp[a] * rx;
    sb += Jp[3 + a] * ry;
    term[9 + a] = wgt * sb;
  }
  ba_accumulate_slot<FRESH>(A, Jx, Jp, wgt, rx, ry);
  return true;
}
// G = Hxp Hpp^-1 and G*bp of one slot (T:1020-1041)
__device__ __forceinline__ void ba_slot_gain(double* __restrict__ A, const double* __restrict__ iH, const double* __restrict__ bp) {
  const double* Hxp = A + 42;
  double* G = A + 60;
  double* gb = A + 78;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const double h0 = Hxp[r * 3 + 0], h1 = Hxp[r * 3 + 1], h2 = Hxp[r * 3 + 2];
    const double g0 = h0 * iH[0] + h1 * iH[3] + h2 * iH[6];
    const double g1 = h0 * iH[1] + h1 * iH[4] + h2 * iH[7];
    const double g2 = h0 * iH[2] + h1 * iH[5] + h2 * iH[8];
    G[r * 3 + 0] = g0; G[r * 3 + 1] = g1; G[r * 3 + 2] = g2;
    gb[r] = g0 * bp[0] + g1 * bp[1] + g2 * bp[2];
  }
}

// per-point records (T:894-1047) of point p by ONE lane: slots in first-observation order, Hxx | bx | Hxp, then G = Hxp Hpp^-1
// and G*bp
__device__ __forceinline__ void ba_point_record(int p, int W, int MS, const double* __restrict__ sp, const double* __restrict__ X,
                                                const int32_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs_li,
                                                const double* __restrict__ obs_uv, double fx, double fy, double cx, double cy, double huber,
                                                double* __restrict__ rec, int8_t* __restrict__ slot_of) {
  int8_t* so = slot_of + (size_t)p * W;
  for (int i = 0; i < W; i++) so[i] = -1;
  const int o0 = obs_ptr[p], o1 = obs_ptr[p + 1];
  if (o1 - o0 > BA_MAX_OBS) return;  // T:915-918
  double* prec = rec + (size_t)p * MS * BA_SLOT;
  const double Xx = X[3 * p], Xy = X[3 * p + 1], Xz = X[3 * p + 2];
  double Hpp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bp[3] = {0, 0, 0};
  int na = 0;
  for (int o = o0; o < o1; ++o) {
    const int li = obs_li[o];
    if (li < 0 || li >= W) continue;  // malformed input: ignore (cannot occur through the host API)
    int ai = so[li];
    const bool fresh = ai < 0;
    if (fresh) {
      ai = na++;
      so[li] = (int8_t)ai;
    }
    double* A = prec + (size_t)ai * BA_SLOT;
    double term[12];
    const bool adds = fresh ? ba_observation<true>(A, sp + 12 * li, Xx, Xy, Xz, obs_uv[2 * o], obs_uv[2 * o + 1], fx, fy, cx, cy, huber, term)
                            : ba_observation<false>(A, sp + 12 * li, Xx, Xy, Xz, obs_uv[2 * o], obs_uv[2 * o + 1], fx, fy, cx, cy, huber, term);
    if (adds) {
#pragma unroll
      for (int e = 0; e < 9; e++) Hpp[e] += term[e];
#pragma unroll
      for (int a = 0; a < 3; a++) bp[a] += term[9 + a];
    }
  }
  double iH[9];
  if (!inv3_ref(Hpp, iH)) {  // T:1012: the point contributes nothing at all
    for (int i = 0; i < W; i++) so[i] = -1;
    return;
  }
  for (int k = 0; k < na; k++) ba_slot_gain(prec + (size_t)k * BA_SLOT, iH, bp);
}

// The same records for PTS points by 16 * PTS lanes of one wave (PTS <= 4): the observations of a point are independent
// up to the ordered sums Hpp / bp, which 12 lanes per point run over the per-observation terms in observation order --
// the reference's `+=` sequence.  A pose that observes a point twice (read-modify-write of its slot) sends that point
// down the one-lane path.  Must be called by all threads of the workgroup (barriers).
template <int PTS>
__device__ __forceinline__ void ba_point_records_wave(int p0, int P, int W, int MS, const double* __restrict__ sp, const double* __restrict__ X,
                                                      const int32_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs_li,
                                                      const double* __restrict__ obs_uv, double fx, double fy, double cx, double cy, double huber,
                                                      double* __restrict__ rec, int8_t* __restrict__ slot_of) {
  static_assert(PTS * BA_MAX_OBS <= 64, "one wave");
  __shared__ double s_term[PTS][BA_MAX_OBS][12];
  __shared__ double s_H[PTS][12], s_iH[PTS][9];
  __shared__ int s_n[PTS], s_na[PTS], s_par[PTS], s_ok[PTS];  // s_par: the point takes the lane-parallel path
  __shared__ int8_t s_slot[PTS][BA_MAX_OBS], s_adds[PTS][BA_MAX_OBS];
  const int tid = threadIdx.x;
  if (tid < PTS) {  // ---- slots in first-observation order (integer work only)
    const int p = p0 + tid;
    int n = 0, na = 0, par = 0;
    if (p < P) {
      int8_t* so = slot_of + (size_t)p * W;
      for (int i = 0; i < W; i++) so[i] = -1;
      const int o0 = obs_ptr[p], cnt = obs_ptr[p + 1] - o0;
      if (cnt <= BA_MAX_OBS) {  // T:915-918
        n = cnt;
        par = 1;
        for (int k = 0; k < cnt; k++) {
          const int li = obs_li[o0 + k];
          int sl = -1;
          if (li >= 0 && li < W) {
            if (so[li] < 0) { sl = na++; so[li] = (int8_t)sl; }
            else par = 0;  // the same pose again
          }
          s_slot[tid][k] = (int8_t)sl;
        }
        if (!par) {
          ba_point_record(p, W, MS, sp, X, obs_ptr, obs_li, obs_uv, fx, fy, cx, cy, huber, rec, slot_of);
          n = 0;
        }
      }
    }
    s_n[tid] = n; s_na[tid] = na; s_par[tid] = par;
  }
  __syncthreads();
  const int pt = tid / BA_MAX_OBS, k = tid % BA_MAX_OBS;
  if (tid < PTS * BA_MAX_OBS && k < s_n[pt]) {  // ---- one lane per observation
    const int p = p0 + pt, sl = s_slot[pt][k];
    bool adds = false;
    if (sl >= 0) {
      const int o = obs_ptr[p] + k;
      adds = ba_observation<true>(rec + ((size_t)p * MS + sl) * BA_SLOT, sp + 12 * obs_li[o], X[3 * p], X[3 * p + 1], X[3 * p + 2], obs_uv[2 * o],
                                  obs_uv[2 * o + 1], fx, fy, cx, cy, huber, s_term[pt][k]);
    }
    s_adds[pt][k] = adds ? 1 : 0;
  }
  __syncthreads();
  if (tid < PTS * 12) {  // ---- Hpp (9) and bp (3): the reference's += chain over the observations
    const int q = tid / 12, e = tid % 12, n = s_n[q];
    double acc = 0.0;
    for (int kk = 0; kk < n; kk++)
      if (s_adds[q][kk]) acc += s_term[q][kk][e];
    s_H[q][e] = acc;
  }
  __syncthreads();
  if (tid < PTS) {
    int ok = 0;
    if (s_n[tid] > 0 || (s_par[tid] && p0 + tid < P)) {
      ok = inv3_ref(s_H[tid], s_iH[tid]) ? 1 : 0;
      if (!ok && s_par[tid]) {  // T:1012: the point contributes nothing at all
        int8_t* so = slot_of + (size_t)(p0 + tid) * W;
        for (int i = 0; i < W; i++) so[i] = -1;
      }
    }
    s_ok[tid] = ok && s_par[tid];
  }
  __syncthreads();
  if (tid < PTS * BA_MAX_OBS && s_ok[pt] && k < s_na[pt])  // ---- one lane per slot
    ba_slot_gain(rec + ((size_t)(p0 + pt) * MS + k) * BA_SLOT, s_iH[pt], s_H[pt] + 9);
}

__device__ __forceinline__ int ba_row_stride(int W) { return 36 * W * W + 36 * W + 12 * W; }

// k_ba_points: BA_PTS points per workgroup.  Phase 1: one lane per point writes the point's records (a serial chain of
// dependent FP64 operations in the reference's order).  Phase 2: all 256 threads spread the records of these points into
// their contribution rows C[p][.] (layout of one row, CS = D*D + 36 W + 2 D doubles):
//   [0, D*D)              Schur term  G_a . Hxp_b  of S element (i,j)   (T:1049-1055), +0.0 if a pose misses the point
//   [D*D, D*D+36W)        Hxx term of the diagonal block of pose A         (T:1017)
//   [.., +D) and [.., +D) bx (T:1018) and G*bp (T:1039-1041)
// A missing contribution is stored as +0.0: the running sums of k_ba_reduce start at +0.0 and can never become -0.0,
// so adding +0.0 is the identity and the reduction needs no branches.  (The expansion used to be a launch of its own.)
// Two shapes (kernels k_ba_points_window / k_ba_points_bulk below): <4, 256> with the expansion (windows of a few hundred
// points: one launch less on the BA chain), and <64, 64> without it (C == nullptr) followed by k_ba_expand over the whole
// device (tens of thousands of points, C4).
#define BA_PTS 4
#define BA_MERGED_EXPAND_MAX_P 4096
template <int PTS, int NT>
__device__ __forceinline__ void ba_points_body(int W, int P, int MS, const double* __restrict__ poses, const double* __restrict__ X,
                                               const int32_t* __restrict__ obs_ptr, const int32_t* __restrict__ obs_li,
                                               const double* __restrict__ obs_uv, double fx, double fy, double cx, double cy,
                                               double huber, double* __restrict__ rec, int8_t* __restrict__ slot_of, double* __restrict__ C,
                                               int wave_prio, int blk) {
  // the BA iterations are the longest dependent chain of a keyframe: their waves go first where they share a SIMD with
  // the bulk kernels of the other lanes (KLT, hypotheses, corner sweeps)
  if (wave_prio) __builtin_amdgcn_s_setprio(3);
  __shared__ double sp[BA_MAX_W * 12];
  const int tid = threadIdx.x;
  __syncthreads();  // (a caller that loops over blocks: the previous block's readers of sp are done)
  for (int i = tid; i < W * 12; i += NT) sp[i] = poses[i];
  __syncthreads();
  const int p0 = blk * PTS;
  if constexpr (PTS * BA_MAX_OBS <= 64 && NT >= 64) {
    ba_point_records_wave<PTS>(p0, P, W, MS, sp, X, obs_ptr, obs_li, obs_uv, fx, fy, cx, cy, huber, rec, slot_of);
  } else {
    if (tid < PTS && p0 + tid < P) ba_point_record(p0 + tid, W, MS, sp, X, obs_ptr, obs_li, obs_uv, fx, fy, cx, cy, huber, rec, slot_of);
  }
  if (C == nullptr) return;
  __syncthreads();  // the records and slot tables of this workgroup's points are visible to all its threads
  const int D = 6 * W, CS = ba_row_stride(W);
  const int np = min(PTS, P - p0);
  for (int e = tid; e < CS; e += NT) {
    // which slot(s) and which doubles inside them feed row element e (the same for every point)
    int pa, pb = -1, off_a, off_b = 0;
    if (e < D * D) {
      const int i = e / D, j = e % D;
      pa = i / 6; pb = j / 6;
      off_a = 60 + (i % 6) * 3; off_b = 42 + (j % 6) * 3;
    } else if (e < D * D + 36 * W) {
      const int k = e - D * D;
      pa = k / 36; off_a = k % 36;
    } else if (e < D * D + 36 * W + D) {
      const int i = e - (D * D + 36 * W);
      pa = i / 6; off_a = 36 + (i % 6);
    } else {
      const int i = e - (D * D + 36 * W + D);
      pa = i / 6; off_a = 78 + (i % 6);
    }
    for (int pl = 0; pl < np; pl++) {
      const int p = p0 + pl;
      const int8_t* so = slot_of + (size_t)p * W;
      const double* base = rec + (size_t)p * MS * BA_SLOT;
      const int sa = so[pa];
      double v = 0.0;
      if (pb >= 0) {
        const int sb = so[pb];
        if (sa >= 0 && sb >= 0) {
          const double* g = base + (size_t)sa * BA_SLOT + off_a;
          const double* h = base + (size_t)sb * BA_SLOT + off_b;
          v = g[0] * h[0] + g[1] * h[1] + g[2] * h[2];
        }
      } else if (sa >= 0) {
        v = base[(size_t)sa * BA_SLOT + off_a];
      }
      C[(size_t)p * CS + e] = v;
    }
  }
}

#define BA_POINTS_PARAMS                                                                                                             \
  int W, int P, int MS, const double *__restrict__ poses, const double *__restrict__ X, const int32_t *__restrict__ obs_ptr,              \
      const int32_t *__restrict__ obs_li, const double *__restrict__ obs_uv, double fx, double fy, double cx, double cy, double huber,   \
      double *__restrict__ rec, int8_t *__restrict__ slot_of, double *__restrict__ C, int wave_prio
#define BA_POINTS_PASS W, P, MS, poses, X, obs_ptr, obs_li, obs_uv, fx, fy, cx, cy, huber, rec, slot_of, C, wave_prio, (int)blockIdx.x
// The window-sized shape is held to 96 VGPRs (the compiler takes 200 when left alone, 400 B of spills at 96 cost nothing
// measurable): its four-wave workgroups need room on all four SIMDs of a CU at once, and next to KLT waves (173 VGPRs
// each, one or two per SIMD) a 200-VGPR wave often finds none -- every launch slower than 60 us in the kernel trace
// overlapped a KLT launch.  In the pipeline 37 -> 26 us per launch (9.5 us alone either way).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_ba_points_window(BA_POINTS_PARAMS) {
  WGS_BEGIN;
  ba_points_body<BA_PTS, 256>(BA_POINTS_PASS);
  WGS_END(1, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(64) void k_ba_points_bulk(BA_POINTS_PARAMS) { ba_points_body<64, 64>(BA_POINTS_PASS); }

// The window shape with everything a workgroup touches between its inputs and its contribution rows in LDS: the observation
// lists, the slot tables and the slot records of its BA_PTS points.  k_ba_points_window keeps the records and the slot tables in
// global memory (they are an output of the bulk shape, which the expansion kernel reads back); for a window the merged expansion
// is their only reader, and by workgroup timestamps (tools/ba_wgstamps.py) a workgroup of that kernel spends 19.6 us -- with the
// device to itself -- almost entirely in L2 round trips: the serial slot assignment (a load per observation, and a load after a
// store on the slot table), the record stores / loads around the gain, and two dependent loads per row entry in the expansion.
// Same arithmetic, same order: ba_observation / the ordered Hpp | bp sums / inv3_ref / ba_slot_gain, unchanged.
// Precondition (checked on the host when the problem is set up): no pose observes a point twice (that case is a read-modify-write
// of the slot record, which stays with the general kernel); points with more than BA_MAX_OBS observations contribute nothing
// (T:915-918).  WT: window size as a compile-time constant (0 = read W), so that the row-entry decoding divides by constants.
template <int WT, int PTS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_ba_points_window_lds(BA_POINTS_PARAMS) {
  WGS_BEGIN;
  static_assert(PTS * BA_MAX_OBS <= 64, "one wave runs the per-observation phases");
  extern __shared__ __align__(16) double s_rec[];  // [PTS][MS][BA_SLOT]
  constexpr int WCAP = WT ? WT : BA_MAX_W;         // (a known window size keeps the static LDS at what it needs: 5.5 KB less at W = 6)
  __shared__ double sp[WCAP * 12];
  __shared__ double s_term[PTS][BA_MAX_OBS][12];
  __shared__ double s_H[PTS][12], s_iH[PTS][9];
  __shared__ int s_n[PTS], s_na[PTS], s_ok[PTS], s_o0[PTS];
  __shared__ int s_li[PTS][BA_MAX_OBS];
  __shared__ int8_t s_slot[PTS][BA_MAX_OBS], s_adds[PTS][BA_MAX_OBS], s_so[PTS][WCAP];
  if (wave_prio) __builtin_amdgcn_s_setprio(3);
  const int Wc = WT ? WT : W;
  const int tid = threadIdx.x, p0 = (int)blockIdx.x * PTS;
  const int pt = tid / BA_MAX_OBS, k = tid % BA_MAX_OBS;
  // ---- inputs: poses, the observation lists of the points (one lane per observation), empty slot tables
  for (int i = tid; i < Wc * 12; i += 256) sp[i] = poses[i];
  for (int i = tid; i < PTS * Wc; i += 256) s_so[i / Wc][i % Wc] = -1;
  double ob_u = 0.0, ob_v = 0.0, pt_x = 0.0, pt_y = 0.0, pt_z = 0.0;  // this lane's observation and its point, fetched with the lists
  if (tid < PTS * BA_MAX_OBS) {
    const int p = p0 + pt;
    int o0 = 0, cnt = 0;
    if (p < P) {
      o0 = obs_ptr[p];
      cnt = obs_ptr[p + 1] - o0;
      pt_x = X[3 * p]; pt_y = X[3 * p + 1]; pt_z = X[3 * p + 2];
    }
    if (cnt > BA_MAX_OBS) cnt = 0;  // T:915-918: the point is skipped as a whole
    if (k == 0) { s_o0[pt] = o0; s_n[pt] = cnt; }
    int li = -1;
    if (k < cnt) {
      li = obs_li[o0 + k];
      ob_u = obs_uv[2 * (o0 + k)];
      ob_v = obs_uv[2 * (o0 + k) + 1];
    }
    s_li[pt][k] = li;
  }
  __syncthreads();
  WGS_MARK(0);  // inputs staged
  // ---- slots in first-observation order (T:925-930): integer work on LDS, one lane per point
  if (tid < PTS) {
    const int n = s_n[tid];
    int na = 0;
    for (int kk = 0; kk < n; kk++) {
      const int li = s_li[tid][kk];
      int sl = -1;
      if (li >= 0 && li < Wc) {  // (malformed pose index: ignored, as in the general kernel)
        sl = na++;               // precondition: this pose has no slot yet
        s_so[tid][li] = (int8_t)sl;
      }
      s_slot[tid][kk] = (int8_t)sl;
    }
    s_na[tid] = na;
  }
  __syncthreads();
  // ---- one lane per observation: the slot record (Hxx | bx | Hxp) and the twelve terms of Hpp | bp
  if (tid < PTS * BA_MAX_OBS && k < s_n[pt]) {
    const int sl = s_slot[pt][k];
    bool adds = false;
    if (sl >= 0)
      adds = ba_observation<true>(s_rec + ((size_t)pt * MS + sl) * BA_SLOT, sp + 12 * s_li[pt][k], pt_x, pt_y, pt_z, ob_u, ob_v, fx, fy, cx, cy, huber,
                                  s_term[pt][k]);
    s_adds[pt][k] = adds ? 1 : 0;
  }
  __syncthreads();
  WGS_MARK(1);  // observations
  if (tid < PTS * 12) {  // ---- Hpp (9) and bp (3): the reference's += chain over the observations
    const int q = tid / 12, e = tid % 12, n = s_n[q];
    double acc = 0.0;
    for (int kk = 0; kk < n; kk++)
      if (s_adds[q][kk]) acc += s_term[q][kk][e];
    s_H[q][e] = acc;
  }
  __syncthreads();
  if (tid < PTS) {
    int ok = 0;
    if (p0 + tid < P && s_n[tid] > 0) ok = inv3_ref(s_H[tid], s_iH[tid]) ? 1 : 0;
    if (!ok)  // T:1012 (or no usable observation): the point contributes nothing at all
      for (int i = 0; i < Wc; i++) s_so[tid][i] = -1;
    s_ok[tid] = ok;
  }
  __syncthreads();
  if (tid < PTS * BA_MAX_OBS && s_ok[pt] && k < s_na[pt])  // ---- one lane per slot: G = Hxp Hpp^-1, G*bp
    ba_slot_gain(s_rec + ((size_t)pt * MS + k) * BA_SLOT, s_iH[pt], s_H[pt] + 9);
  __syncthreads();
  WGS_MARK(2);  // sums, inverse, gain
  // ---- contribution rows (layout: see ba_points_body), every operand from LDS
  const int D = 6 * Wc, CS = ba_row_stride(Wc);
  const int np = min(PTS, P - p0);
  for (int e = tid; e < CS; e += 256) {
    int pa, pb = -1, off_a, off_b = 0;
    if (e < D * D) {
      const int i = e / D, j = e % D;
      pa = i / 6; pb = j / 6;
      off_a = 60 + (i % 6) * 3; off_b = 42 + (j % 6) * 3;
    } else if (e < D * D + 36 * Wc) {
      const int kk = e - D * D;
      pa = kk / 36; off_a = kk % 36;
    } else if (e < D * D + 36 * Wc + D) {
      const int i = e - (D * D + 36 * Wc);
      pa = i / 6; off_a = 36 + (i % 6);
    } else {
      const int i = e - (D * D + 36 * Wc + D);
      pa = i / 6; off_a = 78 + (i % 6);
    }
    // branch-free and batched over the workgroup's points: all slot lookups, then all record reads (a missing slot reads slot 0,
    // a valid address, and is masked out afterwards), then the arithmetic and the stores -- two LDS round trips per row entry
    // instead of up to three per point (the lanes of a wave hold different entries and points with different slot tables, so every
    // branch of the straightforward form was taken by somebody)
    const bool schur = pb >= 0;
    int sa[PTS], sb[PTS];
#pragma unroll
    for (int pl = 0; pl < PTS; pl++) {
      sa[pl] = s_so[pl][pa];
      sb[pl] = schur ? s_so[pl][pb] : 0;
    }
    double g0[PTS], g1[PTS], g2[PTS], h0[PTS], h1[PTS], h2[PTS];
#pragma unroll
    for (int pl = 0; pl < PTS; pl++) {
      const double* base = s_rec + (size_t)pl * MS * BA_SLOT;
      const double* g = base + (size_t)(sa[pl] >= 0 ? sa[pl] : 0) * BA_SLOT + off_a;
      const double* h = base + (size_t)(sb[pl] >= 0 ? sb[pl] : 0) * BA_SLOT + off_b;
      // (unconditional reads: for a one-operand entry the extra five values are read and never used; the dynamic allocation is
      // two doubles longer than the records so that even the last slot's G*bp entry reads inside it)
      g0[pl] = g[0]; g1[pl] = g[1]; g2[pl] = g[2];
      h0[pl] = h[0]; h1[pl] = h[1]; h2[pl] = h[2];
    }
#pragma unroll
    for (int pl = 0; pl < PTS; pl++) {
      const double dot = g0[pl] * h0[pl] + g1[pl] * h1[pl] + g2[pl] * h2[pl];
      const bool have = schur ? (sa[pl] >= 0 && sb[pl] >= 0) : sa[pl] >= 0;
      const double v = have ? (schur ? dot : g0[pl]) : 0.0;
      if (pl < np) C[(size_t)(p0 + pl) * CS + e] = v;
    }
  }
  WGS_MARK(3);  // rows issued
  WGS_END(1, (int)blockIdx.x, (int)gridDim.x);
}
#undef BA_POINTS_PARAMS
#undef BA_POINTS_PASS

// the expansion as a launch of its own (large problems): one workgroup per point; the point's slot records and slot table
// are staged in LDS once and every thread forms 1/256 of the row from there (one thread per element, each fetching its own
// six operands from L2, ran at 1.45 TB/s of stores; the row is 32 KB at W = 10)
// e_lo / e_hi (element-sharded step): only the row entries that feed elements [e_lo, e_hi) of S | b are formed and written --
// the rank that reduces those elements reads nothing else (entry k < D*D feeds S element k; an Hxx entry feeds the diagonal-block
// element at the same (row, column); bx and G*bp entries feed b).
__global__ __launch_bounds__(256) void k_ba_expand(int W, int P, int MS, const double* __restrict__ rec, const int8_t* __restrict__ slot_of,
                                                   double* __restrict__ C, int e_lo, int e_hi, int p_base) {
  // p_base: the launch covers points p_base .. p_base + gridDim.x - 1 and writes their rows to C[0 ..] (a chunk of the row ring)
  __shared__ double srec[BA_MAX_OBS * BA_SLOT];
  __shared__ int8_t sso[BA_MAX_W];
  const int D = 6 * W, CS = ba_row_stride(W);
  const int p = (int)blockIdx.x + p_base, tid = threadIdx.x;
  const double* base = rec + (size_t)p * MS * BA_SLOT;
  for (int k = tid; k < MS * BA_SLOT; k += 256) srec[k] = base[k];
  if (tid < W) sso[tid] = slot_of[(size_t)p * W + tid];
  __syncthreads();
  double* row = C + (size_t)blockIdx.x * CS;
  const bool all = e_lo <= 0 && e_hi >= D * D + D;
  for (int e = tid; e < CS; e += 256) {
    double v = 0.0;
    int feeds;  // the element of S | b this entry is an addend of
    if (e < D * D) {
      feeds = e;
      if (!all && (feeds < e_lo || feeds >= e_hi)) continue;
      const int i = e / D, j = e - i * D;
      const int sa = sso[i / 6], sb = sso[j / 6];
      if (sa >= 0 && sb >= 0) {
        const double* g = srec + sa * BA_SLOT + 60 + (i % 6) * 3;
        const double* h = srec + sb * BA_SLOT + 42 + (j % 6) * 3;
        v = g[0] * h[0] + g[1] * h[1] + g[2] * h[2];
      }
    } else if (e < D * D + 36 * W) {
      const int k = e - D * D;
      const int a = k / 36, rc = k % 36;
      feeds = (6 * a + rc / 6) * D + 6 * a + rc % 6;
      if (!all && (feeds < e_lo || feeds >= e_hi)) continue;
      const int sa = sso[a];
      if (sa >= 0) v = srec[sa * BA_SLOT + rc];
    } else if (e < D * D + 36 * W + D) {
      const int i = e - (D * D + 36 * W);
      feeds = D * D + i;
      if (!all && (feeds < e_lo || feeds >= e_hi)) continue;
      const int sa = sso[i / 6];
      if (sa >= 0) v = srec[sa * BA_SLOT + 36 + (i % 6)];
    } else {
      const int i = e - (D * D + 36 * W + D);
      feeds = D * D + i;
      if (!all && (feeds < e_lo || feeds >= e_hi)) continue;
      const int sa = sso[i / 6];
      if (sa >= 0) v = srec[sa * BA_SLOT + 78 + (i % 6)];
    }
    row[e] = v;
  }
}

// ------------------------------------------------------------------------------------------ dense solve
// cpp/include/dense.hpp:54-93 in the reference's operation order.  status[0] = 0 ok, 1 singular
// (pivot < 1e-15, where the reference throws).  A,b are read from global memory, x written.
//
// k_solve_wave (n <= 64): ONE wavefront, matrix in LDS, lane = row.  Nothing but the wave's own
// in-order LDS traffic synchronises the phases (block = 1 wave, so __syncthreads() is free):
//   pivot   : lane i reads |A[i][k]|, wave max by xor-shuffles, first lane holding the max = the
//             reference's "first strictly larger" scan; NaN never wins unless it sits on the diagonal;
//   swap    : lanes as columns exchange rows k and piv (columns k..n-1 only, like the reference);
//   scale   : A[k][j] /= akk (IEEE division), b[k] /= akk;
//   eliminate: lane i (row i > k, |f_i| >= 1e-18) walks its row: A[i][j] -= f_i * A[k][j]; the row-k
//             operands are LDS broadcasts, the own-row accesses are conflict-free (odd row stride);
//   back-substitution: products A(i,j)*x_j are formed by all rows as soon as x_j exists, so only the
//             reference's ascending-j subtraction chain of the current row is serial.
#define SOLVE_WAVE_MAX_N 64
// wave-wide maximum of non-NaN doubles by DPP (no LDS round trips): after the row steps every lane of a 16-lane row
// holds the row maximum; row_bcast15 / row_bcast31 carry it across rows; lane 63 ends up with the wave maximum.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double m) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(m), __double2loint(m), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(m), __double2hiint(m), CTRL, ROW_MASK, 0xF, false);
  const double t = __hiloint2double(hi, lo);
  return t > m ? t : m;
}
__device__ __forceinline__ double wave_max_f64(double m) {
  m = dpp_max_step<0xB1, 0xF>(m);   // quad_perm [1,0,3,2]
  m = dpp_max_step<0x4E, 0xF>(m);   // quad_perm [2,3,0,1]
  m = dpp_max_step<0x141, 0xF>(m);  // row_half_mirror
  m = dpp_max_step<0x140, 0xF>(m);  // row_mirror
  m = dpp_max_step<0x142, 0xA>(m);  // row_bcast15 into rows 1 and 3
  m = dpp_max_step<0x143, 0xC>(m);  // row_bcast31 into rows 2 and 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(m), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(m), 63);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void k_solve_wave(const double* __restrict__ Ain, const double* __restrict__ bin, int n,
                                                   double* __restrict__ x, int* __restrict__ status) {
  __shared__ double A[SOLVE_WAVE_MAX_N * (SOLVE_WAVE_MAX_N + 1)];
  __shared__ double bb[SOLVE_WAVE_MAX_N];
  __shared__ double xs[SOLVE_WAVE_MAX_N];
  const int lane = threadIdx.x;
  const int ld = n | 1;  // odd row stride
  for (int e = lane; e < n * n; e += 64) A[(e / n) * ld + (e % n)] = Ain[e];
  if (lane < n) bb[lane] = bin[lane];
  __syncthreads();
  for (int k = 0; k < n; k++) {
    // ---- pivot (dense.hpp:61-67)
    const bool mine = lane >= k && lane < n;
    const double v = mine ? fabs(A[lane * ld + k]) : -1.0;
    const double akk0 = fabs(A[k * ld + k]);
    const double m = wave_max_f64((v == v) ? v : -1.0);  // NaN below the diagonal never wins `v > best`
    int piv = k;
    double best = akk0;
    if (akk0 == akk0) {  // a NaN on the diagonal stays the pivot: `v > NaN` is never true
      const unsigned long long hit = __ballot(mine && v == m);
      piv = hit ? (int)__builtin_ctzll(hit) : k;
      best = m;
    }
    if (best < 1e-15) {
      if (lane == 0) status[0] = 1;
      return;
    }
    // ---- row swap on columns k..n-1 (lanes as columns) fused with the normalisation of the new row k:
    // A[k][j] = A[piv][j] / A[piv][k], A[piv][j] = old A[k][j]; b likewise.  Every lane reads its operands before
    // it stores (the stores may alias the loads, so the compiler keeps that order), and the wave runs in lockstep.
    const int j = k + lane;
    const double akk = A[piv * ld + k];
    {
      // b rides along as "column n" (lane n-k), so that its division shares the instruction with the row's
      const bool b_lane = j == n;
      double* pk = b_lane ? &bb[k] : &A[k * ld + j];
      double* pp = b_lane ? &bb[piv] : &A[piv * ld + j];
      if (j <= n) {
        const double top = *pk, low = *pp;
        *pk = low / akk;
        if (piv != k) *pp = top;
      }
      if (n - k > 63 && lane == 0) {  // n == 64, k == 0: there is no lane 64
        const double bk = bb[k], bp = bb[piv];
        bb[k] = bp / akk;
        if (piv != k) bb[piv] = bk;
      }
    }
    __syncthreads();
    // ---- eliminate rows below (dense.hpp:78-83): row r = k+1+(lane / L) is shared by L = 2^s lanes, each taking
    // every L-th column -- the element updates a_rc -= f_r * a_kc are independent of each other.  All lanes of a row
    // read the multiplier f_r = a_rk in the same instruction, before the lane owning column k overwrites it.
    const int rows = n - k - 1;
    if (rows > 0) {
      int L = 1;
      while (2 * L * rows <= 64) L *= 2;
      const int r = k + 1 + lane / L, part = lane % L;
      if (lane / L < rows) {
        double* row = A + r * ld;
        const double* rk = A + k * ld;
        const double f = row[k];
        if (!(fabs(f) < 1e-18)) {
          int c = k + part;
          for (; c + 3 * L < n; c += 4 * L) {
            const double r0 = rk[c], r1 = rk[c + L], r2 = rk[c + 2 * L], r3 = rk[c + 3 * L];
            const double a0 = row[c], a1 = row[c + L], a2 = row[c + 2 * L], a3 = row[c + 3 * L];
            row[c] = a0 - f * r0; row[c + L] = a1 - f * r1; row[c + 2 * L] = a2 - f * r2; row[c + 3 * L] = a3 - f * r3;
          }
          for (; c < n; c += L) row[c] = row[c] - f * rk[c];
          if (part == 0) bb[r] = bb[r] - f * bb[k];
        }
      }
    }
    __syncthreads();
  }
  // ---- back substitution (dense.hpp:86-91)
  for (int jx = n - 1; jx >= 0; jx--) {
    if (lane == 0) {
      double sacc = bb[jx];
      const double* row = A + jx * ld;
      int c = jx + 1;
      for (; c + 8 <= n; c += 8) {
        const double p0 = row[c], p1 = row[c + 1], p2 = row[c + 2], p3 = row[c + 3], p4 = row[c + 4], p5 = row[c + 5], p6 = row[c + 6],
                     p7 = row[c + 7];
        sacc -= p0; sacc -= p1; sacc -= p2; sacc -= p3; sacc -= p4; sacc -= p5; sacc -= p6; sacc -= p7;
      }
      for (; c < n; c++) sacc -= row[c];
      xs[jx] = sacc;
    }
    __syncthreads();
    const double xj = xs[jx];
    if (lane < jx) A[lane * ld + jx] = A[lane * ld + jx] * xj;  // the product the reference forms at dense.hpp:89
    __syncthreads();
  }
  if (lane == 0) status[0] = 0;
  if (lane < n) x[lane] = xs[lane];
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step_i32(int m) {
  const int t = __builtin_amdgcn_update_dpp(m, m, CTRL, ROW_MASK, 0xF, false);
  return t < m ? t : m;
}
__device__ __forceinline__ int wave_min_i32(int m) {
  m = dpp_min_step_i32<0xB1, 0xF>(m);
  m = dpp_min_step_i32<0x4E, 0xF>(m);
  m = dpp_min_step_i32<0x141, 0xF>(m);
  m = dpp_min_step_i32<0x140, 0xF>(m);
  m = dpp_min_step_i32<0x142, 0xA>(m);
  m = dpp_min_step_i32<0x143, 0xC>(m);
  return __builtin_amdgcn_readlane(m, 63);
}
__device__ __forceinline__ double readlane_f64(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// wave-wide maximum of 32-bit unsigned keys (0 = identity, so the DPP move folds into v_max_u32); result in every lane's SGPR copy
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax_step(unsigned m) {
  const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, CTRL, ROW_MASK, 0xF, false);
  return t > m ? t : m;
}
__device__ __forceinline__ unsigned wave_umax_u32(unsigned m) {
  m = dpp_umax_step<0xB1, 0xF>(m);
  m = dpp_umax_step<0x4E, 0xF>(m);
  m = dpp_umax_step<0x141, 0xF>(m);
  m = dpp_umax_step<0x140, 0xF>(m);
  m = dpp_umax_step<0x142, 0xA>(m);
  m = dpp_umax_step<0x143, 0xC>(m);
  return (unsigned)__builtin_amdgcn_readlane((int)m, 63);
}
// one wave: its LDS operations execute in program order, so a compiler-level fence is all the phases need
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// k_solve_regs<N>: ONE wavefront, lane = row, the row's N coefficients + b in registers (static indices: the k loop
// is unrolled).  Rows never move: a row swap exchanges the POSITIONS two lanes stand for.  Per pivot step:
//   pivot     |a_ik| are non-negative, so their order is the order of their bit patterns: two 6-step DPP maxima over
//             32-bit halves (NaN keys are 0: `v > best` is never true for them), first position among equal maxima;
//   normalise the pivot lane writes its raw row to LDS, lane j divides element j by the pivot (one IEEE division per
//             lane) and stores the quotient as row k of R: R ends up holding the normalised rows BY POSITION, which is
//             all the back substitution reads; the pivot lane itself never takes its row back;
//   eliminate every remaining lane: a_ij = a_ij - f_i * R[k][j], operands broadcast from LDS, separate multiply and
//             subtract; nothing else touches the registers (no selects).
// `ws` is N * ((N+1)|1) + N + 2 doubles of LDS.  Callable from wave 0 of any workgroup (no s_barrier inside).
template <int N, bool STAMP = false>
__device__ __forceinline__ void solve_regs_wave(const double* __restrict__ Ain, const double* __restrict__ bin, double* __restrict__ x,
                                                int* __restrict__ status, double* __restrict__ ws, int lane,
                                                unsigned long long* __restrict__ stamps = nullptr, double* x_lane = nullptr,
                                                int* status_out = nullptr) {
  if constexpr (STAMP) { if (lane == 0) stamps[0] = __builtin_amdgcn_s_memtime(); }
  static_assert(N >= 2 && N <= 62, "b rides along as column N: lane N must exist");
  constexpr int LD = (N + 1) | 1;
  double* R = ws;
  double* raw = ws + N * LD;
  const int row = lane < N ? lane : N - 1;  // lanes >= N shadow the last row and never take part
  double a[N];
  {  // every lane fetches its own row: N/2 independent 16-byte loads in flight (rows are 16-byte aligned: N is even)
    static_assert(N % 2 == 0, "16-byte row loads");
    const double2* src = reinterpret_cast<const double2*>(Ain + (size_t)row * N);
#pragma unroll
    for (int j = 0; j < N / 2; j++) {
      const double2 t = src[j];
      a[2 * j] = t.x;
      a[2 * j + 1] = t.y;
    }
  }
  double bv = bin[row];
  if constexpr (STAMP) { if (lane == 0) stamps[1] = __builtin_amdgcn_s_memtime(); }
  int pos = lane < N ? lane : 0x3fffffff;  // position of this lane's row; finished pivot rows keep position k
  bool done = lane >= N;                   // true once the row has been a pivot row
#pragma unroll
  for (int k = 0; k < N; k++) {
    if constexpr (STAMP) { if (lane == 0) stamps[2 + k] = __builtin_amdgcn_s_memtime(); }
    // ---- pivot (dense.hpp:61-67), kept on the vector side: the scalar unit only sees the rare cases
    const bool cand_row = !done;
    const double v = fabs(a[k]);
    const bool at_k = cand_row && pos == k;
    const bool v_nan = v != v;
    bool is_p;   // this lane's row is the pivot row of step k
    int pivpos;  // the position it comes from
    if (__ballot(at_k && v_nan)) {  // a NaN on the diagonal stays the pivot: `v > NaN` is never true, `NaN < 1e-15` neither
      is_p = at_k;
      pivpos = k;
    } else {
      const bool ok = cand_row && !v_nan;
      const unsigned hi = ok ? (unsigned)__double2hiint(v) : 0u;
      const unsigned mh = wave_umax_u32(hi);
      bool hit = ok && hi == mh;  // never empty: the row at position k is ok
      unsigned long long hits = __ballot(hit);
      if (__builtin_popcountll(hits) != 1) {  // rare: the upper halves tie; lower halves, then the first position (the reference's `>`)
        const unsigned lo = hit ? (unsigned)__double2loint(v) : 0u;
        const unsigned ml = wave_umax_u32(lo);
        hit = hit && lo == ml;
        const int first = wave_min_i32(hit ? pos : 0x7fffffff);
        hit = cand_row && pos == first;
        hits = __ballot(hit);
      }
      if (__ballot(hit && v < 1e-15)) {
        if (lane == 0) status[0] = 1;
        if (status_out) *status_out = 1;
        return;
      }
      is_p = hit;
      pivpos = __builtin_amdgcn_readlane(pos, (int)__builtin_ctzll(hits));
    }
    if constexpr (STAMP) { if (lane == 0) stamps[40 + 3 * k] = __builtin_amdgcn_s_memtime(); }
    // ---- row swap (dense.hpp:69-72) = exchange of positions
    if (at_k) pos = pivpos;
    if (is_p) { pos = k; done = true; }
    // ---- normalise the pivot row (dense.hpp:74-76), transposed through LDS: one division per lane.  The copy starts at
    // an even column so that it is made of aligned 16-byte stores for either parity of k (one stale element rides along)
    if (is_p) {
#pragma unroll
      for (int j = (k & ~1); j < N; j++) raw[j] = a[j];
      raw[N] = bv;
    }
    wave_sync();
    if constexpr (STAMP) { if (lane == 0) stamps[41 + 3 * k] = __builtin_amdgcn_s_memtime(); }
    const double akk = raw[k];
    const double q = raw[lane <= k ? k : (lane > N ? N : lane)] / akk;  // lane j: element j of the normalised row (b at lane N)
    if (lane > k && lane <= N) R[k * LD + lane] = q;                  // kept by position for the back substitution
    // ---- eliminate (dense.hpp:78-83); column k itself becomes f - f*1 and is never read again.  The normalised row
    // reaches every lane as scalar operands (v_readlane from the lane that divided it): no LDS round trip, no waits
    if constexpr (STAMP) { if (lane == 0) stamps[42 + 3 * k] = __builtin_amdgcn_s_memtime() + (unsigned long long)(q == 123.0); }
    const double f = a[k];
    if (!done && !(fabs(f) < 1e-18)) {
#pragma unroll
      for (int j = k + 1; j < N; j++) a[j] = a[j] - f * readlane_f64(q, j);
      bv = bv - f * readlane_f64(q, N);
    }
  }
  // ---- back substitution (dense.hpp:86-91): lane i now reads the normalised row of POSITION i
  wave_sync();
  if constexpr (STAMP) { if (lane == 0) stamps[2 + N] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
  for (int j = 1; j < N; j++) a[j] = R[row * LD + j];  // entries j <= row are stale and never used
  const double s0 = R[row * LD + N];
  double xr = 0.0;
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double s = s0;
#pragma unroll
    for (int j = i + 1; j < N; j++) s -= a[j];  // a[j] already holds A(.,j)*x[j] for this lane's row
    const double xi = readlane_f64(s, i);
    if (i > 0) a[i] = a[i] * xi;  // the product the reference forms at dense.hpp:89, for every row at once
    if (lane == i) xr = xi;
  }
  if (lane == 0) status[0] = 0;
  if (lane < N) x[lane] = xr;
  if (x_lane) *x_lane = xr;
  if (status_out) *status_out = 0;
  if constexpr (STAMP) { if (lane == 0) stamps[3 + N] = __builtin_amdgcn_s_memtime(); }
}
template <int N>
__global__ __launch_bounds__(64) void k_solve_regs(const double* __restrict__ Ain, const double* __restrict__ bin, double* __restrict__ x,
                                                   int* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) double ws[N * ((N + 1) | 1) + N + 2];
  solve_regs_wave<N>(Ain, bin, x, status, ws, (int)threadIdx.x);
}
// diagnostic instantiation (SFMX_SOLVE_STAMPS=1): s_memtime at entry, after the load, before every pivot step, before and
// after the back substitution
template <int N>
__global__ __launch_bounds__(64) void k_solve_regs_stamps(const double* __restrict__ Ain, const double* __restrict__ bin, double* __restrict__ x,
                                                          int* __restrict__ status, unsigned long long* __restrict__ stamps) {
  __shared__ __attribute__((aligned(16))) double ws[N * ((N + 1) | 1) + N + 2];
  solve_regs_wave<N, true>(Ain, bin, x, status, ws, (int)threadIdx.x, stamps);
}

// Ordered column sums of the contribution rows = the reduced camera system S | b (T:1017-1018, 1039-1055).  The add
// chain of one element over the points is strictly serial (reference order), its operands are not: a workgroup owns
// BAR_COLS neighbouring elements, all 256 threads stream the next tile of BAR_TP contribution rows into LDS (independent
// loads, 128-byte segments), two tiles ahead of the 16 lanes of wave 0 that run the add chains.  Missing second addends are
// +0.0 (identity, see k_ba_points); "b -= G*bp" is evaluated as b += (-(G*bp)), which is the same IEEE operation.
// SOLVE_N = 6 W (36, 60): the workgroup that finishes LAST (device-scope ticket) goes on to solve S dx = b with
// solve_regs_wave and publishes dx | status (to `work`, and to pinned host memory with a sequence word when `host_out`
// is given): one BA iteration is two launches (points + expansion, reduction + solve) instead of five.
#define BAR_COLS 16
#define BAR_Q (256 / BAR_COLS)        // point phases per tile pass
// LDS: 2 arrays x 3 tiles x 64 rows x 16 columns x 8 B = 48 KiB per workgroup (a 128 KiB double buffer could not start
// on a CU that still held KLT workgroups, and the kernel took 2-3 x longer inside the pipeline than alone).
// BAR_NPF: tiles whose loads are in flight in registers; BAR_NBUF: LDS tiles in rotation (2 suffice: tile t+2 is stored after
// the barrier that follows the chain over tile t).  Window-sized problems run <64, N, 2, 2>: 32 KiB of LDS and 96 VGPRs -- a
// small footprint finds a CU sooner next to KLT workgroups.  The streaming shape that C4 uses is <128, N, 2, 2>: 64 KiB of
// LDS, half as many barriers per point (C4 reduction 1.14 -> 0.91 ms; <64, N, 4, 3> before).
template <int BAR_TP, int SOLVE_N, int BAR_NPF, int BAR_NBUF, bool LOOKAHEAD = false>
__device__ __forceinline__ void ba_reduce_body(int W, int P, const double* __restrict__ C, double lambda, int damp, double* __restrict__ S,
                                               double* __restrict__ b, unsigned* __restrict__ ticket, double* __restrict__ work,
                                               double* __restrict__ host_out, unsigned long long seq, int wave_prio, const double* init,
                                               int blk, int nblk, int publish_system WGS_PARAM) {
  // blk / nblk: this workgroup's element block and how many blocks the launch has (the ticket of the fused solve counts them)
  // init (optional, [D*D + D] in S | b layout, may alias S): the chains start from these values instead of +0.0 -- a shard
  // that continues the running sums of the shard before it (relay mode: the reference's sequence across shards)
  if (wave_prio) __builtin_amdgcn_s_setprio(3);  // see k_ba_points
  constexpr int BAR_K = BAR_TP / BAR_Q;  // rows per thread and tile
  constexpr int TILE_DOUBLES = 2 * BAR_NBUF * BAR_TP * BAR_COLS;
  constexpr int SOLVE_DOUBLES = SOLVE_N > 0 ? SOLVE_N * ((SOLVE_N + 1) | 1) + SOLVE_N + 2 : 0;
  __shared__ __attribute__((aligned(16))) double lds[TILE_DOUBLES > SOLVE_DOUBLES ? TILE_DOUBLES : SOLVE_DOUBLES];
  double (*sv)[BAR_TP][BAR_COLS] = reinterpret_cast<double (*)[BAR_TP][BAR_COLS]>(lds);
  double (*su)[BAR_TP][BAR_COLS] = reinterpret_cast<double (*)[BAR_TP][BAR_COLS]>(lds + BAR_NBUF * BAR_TP * BAR_COLS);
  const int D = 6 * W, CS = ba_row_stride(W), NE = D * D + D;
  const int tid = threadIdx.x, col = tid % BAR_COLS, q = tid / BAR_COLS;
  const int e_raw = blk * BAR_COLS + col;
  const bool valid = e_raw < NE;
  const int e = valid ? e_raw : NE - 1;
  const bool is_b = e >= D * D;
  const int i = is_b ? e - D * D : e / D;
  const int j = is_b ? 0 : e % D;
  const bool diag_blk = (!is_b) && (i / 6 == j / 6);
  // first / second addend of this element inside a contribution row
  const int o1 = is_b ? (D * D + 36 * W + i) : (D * D + (i / 6) * 36 + (i % 6) * 6 + (j % 6));
  const int o2 = is_b ? (D * D + 36 * W + D + i) : e;
  const bool two = is_b || diag_blk;
  const bool any_two = __any(two);  // uniform over the block: every wave holds the same BAR_COLS columns
  const int ntiles = (P + BAR_TP - 1) / BAR_TP;
  // Register ring of BAR_NPF tiles: the loads of tile t + BAR_NPF are issued when tile t is parked in LDS, i.e. BAR_NPF
  // chain durations (~0.6 us each) before they are needed -- more than the ~2 us a load takes under load.  (With a
  // look-ahead of one tile every iteration waited ~1 us for its loads: 24 us for the 10 tiles of a 600-point window.)
  double rv[BAR_NPF][BAR_K], ru[BAR_NPF][BAR_K];
  auto load_tile = [&](int t, double (&pv)[BAR_K], double (&pu)[BAR_K]) {
#pragma unroll
    for (int k = 0; k < BAR_K; k++) {
      const int p = min(t * BAR_TP + k * BAR_Q + q, P - 1);
      const double* row = C + (size_t)p * CS;
      const double v = row[o2];
      pv[k] = is_b ? -v : v;
      pu[k] = two ? row[o1] : 0.0;
    }
  };
  auto store_tile = [&](int buf, const double (&pv)[BAR_K], const double (&pu)[BAR_K]) {
#pragma unroll
    for (int k = 0; k < BAR_K; k++) {
      sv[buf][k * BAR_Q + q][col] = pv[k];
      su[buf][k * BAR_Q + q][col] = pu[k];
    }
  };
  double acc = (init && tid < BAR_COLS) ? init[e] : 0.0;
#pragma unroll
  for (int j = 0; j < BAR_NPF; j++)
    if (j < ntiles) load_tile(j, rv[j], ru[j]);
  WGS_MARK(0);  // indices, first loads issued
  for (int t0 = 0; t0 < ntiles; t0 += BAR_NPF) {
#pragma unroll
    for (int j = 0; j < BAR_NPF; j++) {
      const int t = t0 + j;
      if (t < ntiles) {  // uniform
        const int buf = t % BAR_NBUF;  // held tile t - BAR_NBUF, whose chain finished before the barrier of the tile after it
        store_tile(buf, rv[j], ru[j]);
        if (t + BAR_NPF < ntiles) load_tile(t + BAR_NPF, rv[j], ru[j]);
        __syncthreads();
        if (tid < BAR_COLS) {
          // the LDS reads of the next batch of rows are issued before the dependent adds of the current one
          const int cnt = min(BAR_TP, P - t * BAR_TP);
          // (LOOKAHEAD for the window shape was measured with workgroup timestamps: chains done at 12.9 us against 13.1 -- the adds
          // themselves are the time; not instantiated)
          if (cnt == BAR_TP && BAR_NPF <= 2 && BAR_TP <= 64 && !LOOKAHEAD) {  // one batch in registers (32 VGPRs less than the look-ahead below)
            if (any_two) {
#pragma unroll
              for (int bch = 0; bch < BAR_TP / 8; bch++) {
                double u[8], v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { u[k] = su[buf][bch * 8 + k][col]; v[k] = sv[buf][bch * 8 + k][col]; }
#pragma unroll
                for (int k = 0; k < 8; k++) { acc += u[k]; acc += v[k]; }
              }
            } else {
#pragma unroll
              for (int bch = 0; bch < BAR_TP / 16; bch++) {
                double v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) v[k] = sv[buf][bch * 16 + k][col];
#pragma unroll
                for (int k = 0; k < 16; k++) acc += v[k];
              }
            }
          } else if (cnt == BAR_TP) {  // streaming shape: the LDS reads of the next batch ahead of the adds of the current one
            if (any_two) {
              double u[2][8], v[2][8];
#pragma unroll
              for (int k = 0; k < 8; k++) { u[0][k] = su[buf][k][col]; v[0][k] = sv[buf][k][col]; }
#pragma unroll
              for (int bch = 0; bch < BAR_TP / 8; bch++) {
                if (bch + 1 < BAR_TP / 8) {
#pragma unroll
                  for (int k = 0; k < 8; k++) { u[(bch + 1) & 1][k] = su[buf][(bch + 1) * 8 + k][col]; v[(bch + 1) & 1][k] = sv[buf][(bch + 1) * 8 + k][col]; }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) { acc += u[bch & 1][k]; acc += v[bch & 1][k]; }  // S += Hxx ; S += G_a Hxp_b^T (reference ADDS, Q6) | b += bx ; b -= G*bp
              }
            } else {
              double v[2][16];
#pragma unroll
              for (int k = 0; k < 16; k++) v[0][k] = sv[buf][k][col];
#pragma unroll
              for (int bch = 0; bch < BAR_TP / 16; bch++) {
                if (bch + 1 < BAR_TP / 16) {
#pragma unroll
                  for (int k = 0; k < 16; k++) v[(bch + 1) & 1][k] = sv[buf][(bch + 1) * 16 + k][col];
                }
#pragma unroll
                for (int k = 0; k < 16; k++) acc += v[bch & 1][k];
              }
            }
          } else if (any_two) {  // the last, partial tile
            for (int pp = 0; pp < cnt; pp++) { acc += su[buf][pp][col]; acc += sv[buf][pp][col]; }
          } else {
            for (int pp = 0; pp < cnt; pp++) acc += sv[buf][pp][col];
          }
        }
      }
    }
  }
  WGS_MARK(1);  // chains done
  __syncthreads();  // the LDS tiles are reused by the solve below
  if (tid < BAR_COLS && valid) {
    if (is_b) {
      if (damp && i < 6) acc = 0.0;  // T:1070
      b[i] = acc;
    } else {
      if (damp && i == j) {
        acc += lambda;          // T:1064
        if (i < 6) acc += 1e9;  // T:1069
      }
      S[(size_t)i * D + j] = acc;
    }
  }
  if constexpr (SOLVE_N > 0) {
    __shared__ int is_last;
    if (publish_system == 2) {
      // The host solves (csrc/hip/solve_host.cpp) and every workgroup hands ITS 16 elements of S | b to the pinned block itself
      // (one 128-byte write), made visible system-wide before its ticket; the workgroup that takes the last ticket only writes the
      // sequence word the host polls.  (publish_system == 1: the last workgroup copies all of S | b, 10 KB at 36 unknowns, after the
      // others are done -- 5 us at the end of the launch by workgroup timestamps.)
      if (tid < BAR_COLS && valid) {
        host_out[e] = acc;
        __threadfence_system();
      }
      __syncthreads();
      if (tid == 0) {
        const bool last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nblk - 1u;
        if (last) {
          __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch on this stream
          __threadfence_system();
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + (SOLVE_N * SOLVE_N + SOLVE_N)), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      WGS_MARK(2);
      return;
    }
    // ---- the last workgroup to get here solves the system the others have just finished writing
    if (tid < BAR_COLS) __threadfence();  // this workgroup's elements are visible device-wide before its ticket is
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(ticket, 1u) == (unsigned)nblk - 1u ? 1 : 0;
    __syncthreads();
    WGS_MARK(2);  // elements stored, ticket taken
    if (!is_last) return;
    __threadfence();  // acquire: S | b of every other workgroup
    if (tid == 0) *ticket = 0;  // for the next launch on this stream
    if (publish_system) {
      // the host solves (csrc/hip/solve_host.cpp): S | b -- 10 KB at 36 unknowns -- into pinned host memory by this one workgroup,
      // then the sequence word (system-scope release) the host is polling
      const int ne = SOLVE_N * SOLVE_N + SOLVE_N;
      for (int k = tid; k < ne; k += 256) host_out[k] = S[k];  // S | b are contiguous
      __threadfence_system();
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + ne), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (tid >= 64) return;
    double xr = 0.0;
    int st = 0;
    int* dstatus = reinterpret_cast<int*>(work + SOLVE_N);
    solve_regs_wave<SOLVE_N>(S, b, work, dstatus, lds, tid, nullptr, &xr, &st);
    if (host_out) {  // dx | status | sequence word in pinned host memory: the host polls the word (no DMA copy, no stream sync)
      if (tid < SOLVE_N) host_out[tid] = xr;
      if (tid == 0) *reinterpret_cast<int*>(host_out + SOLVE_N) = st;
      __threadfence_system();
      wave_sync();
      if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + SOLVE_N + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int BAR_TP, int SOLVE_N, int BAR_NPF, int BAR_NBUF, bool LOOKAHEAD = false>
__global__ __launch_bounds__(256) void k_ba_reduce(int W, int P, const double* __restrict__ C, double lambda, int damp,
                                                   double* __restrict__ S, double* __restrict__ b,
                                                   unsigned* __restrict__ ticket, double* __restrict__ work, double* __restrict__ host_out,
                                                   unsigned long long seq, int wave_prio, const double* init = nullptr, int wg_off = 0,
                                                   int publish_system = 0) {
  // wg_off: element-sharded launches cover a slice of the element blocks
  WGS_BEGIN;
  ba_reduce_body<BAR_TP, SOLVE_N, BAR_NPF, BAR_NBUF, LOOKAHEAD>(W, P, C, lambda, damp, S, b, ticket, work, host_out, seq, wave_prio, init,
                                                                (int)blockIdx.x + wg_off, (int)gridDim.x, publish_system WGS_ARG);
#ifdef SFMX_BA_WGSTAMPS
  // (the body returns early in all workgroups but the last of a fused launch: stamped here only when it falls through)
#endif
  WGS_END(2, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------------ resident window BA
// One BA job of the pipeline is `iters` (5) iterations on the same window, each a host round trip (the SO(3) update needs the
// platform libm) -- ten launches whose workgroups queue behind the KLT / hypothesis / corner kernels of the other lanes every
// time: inside the pipeline an iteration took ~80 us against ~25 us with the device to itself.  k_ba_window_resident is
// launched ONCE per job and stays: its workgroups (one per block of 16 elements of S | b, 84 at 36 unknowns) run the points
// phase of an iteration (a few 4-point batches each), meet at a device-wide barrier, reduce their element block over all
// points, the last one publishes S | b into pinned host memory -- and then all of them wait for the host's next command word
// (the iteration number whose poses it has written into the pinned pose block, or EXIT).  The arithmetic is the two kernels'
// (the same device functions); only who waits where changes.
// Every wait is bounded (BAW_SPIN_LIMIT polls, ~0.2 s): a kernel that gives up raises ctl->abort and leaves; the host then
// reports an error instead of hanging, and nothing ever spins on a device that has lost its host.
#define BAW_EXIT 0xffffffffffffffffull
#define BAW_SPIN_LIMIT 400000
struct BaResidentCtl {          // device memory, zeroed before every launch
  unsigned arrive;              // barrier between the points and the reduction phase (monotonic)
  unsigned ticket;              // finished reductions (monotonic)
  unsigned abort;               // a wait ran out
  unsigned pad;
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_ba_window_resident(
    int W, int P, int MS, const double* poses_host, const double* __restrict__ X, const int32_t* __restrict__ obs_ptr,
    const int32_t* __restrict__ obs_li, const double* __restrict__ obs_uv, double fx, double fy, double cx, double cy, double huber,
    double lambda, double* __restrict__ rec, int8_t* __restrict__ slot_of, double* __restrict__ C, double* __restrict__ S,
    double* __restrict__ b, BaResidentCtl* ctl, const unsigned long long* cmd_host, double* result_host, unsigned long long first_seq, int max_iters,
    int wave_prio, int n_elem_blocks) {
  // n_elem_blocks <= gridDim.x: the first n_elem_blocks workgroups reduce one block of 16 elements each; all of them take part in
  // the points phase and in the barriers
  WGS_BEGIN;
  __shared__ unsigned long long s_cmd;
  __shared__ int s_flag;
  __shared__ double s_poses[BA_MAX_W * 12];
  const int tid = threadIdx.x, blk = blockIdx.x, nblk = gridDim.x;
  const int D = 6 * W, NE = D * D + D;
  const int nbatch = (P + BA_PTS - 1) / BA_PTS;
  for (int it = 0; it < max_iters; ++it) {
    const unsigned long long want = first_seq + (unsigned long long)it;
    // ---- the host's command: poses of iteration `want` are in place, or EXIT
    if (tid == 0) {
      unsigned long long v = 0;
      int spins = 0;
      for (;;) {
        v = __hip_atomic_load(cmd_host, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == want || v == BAW_EXIT) break;
        if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++spins > BAW_SPIN_LIMIT) { v = BAW_EXIT; atomicExch(&ctl->abort, 1u); break; }
        __builtin_amdgcn_s_sleep(8);
      }
      s_cmd = v;
    }
    __syncthreads();
    if (s_cmd != want) return;
    // ---- points phase: 4-point batches blk, blk + nblk, ... (records + contribution rows).  The poses change from iteration
    // to iteration inside this one launch: they are fetched with system-scope loads (never from a cache line of the iteration
    // before, never hoisted) into LDS, which is what the batches then read
    for (int i = tid; i < W * 12; i += 256) s_poses[i] = __hip_atomic_load(poses_host + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    for (int bt = blk; bt < nbatch; bt += nblk)
      ba_points_body<BA_PTS, 256>(W, P, MS, s_poses, X, obs_ptr, obs_li, obs_uv, fx, fy, cx, cy, huber, rec, slot_of, C, wave_prio, bt);
    // ---- every row is written before any element block is reduced
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&ctl->arrive, 1u);
      const unsigned target = (unsigned)(it + 1) * (unsigned)nblk;
      int spins = 0, ok = 1;
      while (__hip_atomic_load(&ctl->arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ++spins > BAW_SPIN_LIMIT) { ok = 0; atomicExch(&ctl->abort, 1u); break; }
        __builtin_amdgcn_s_sleep(2);
      }
      s_flag = ok;
    }
    __syncthreads();
    if (!s_flag) return;
    __threadfence();
    // ---- reduction of this workgroup's element block over all points (damping and gauge included, T:1064-1071)
    if (blk < n_elem_blocks) ba_reduce_body<64, 0, 2, 2>(W, P, C, lambda, 1, S, b, nullptr, nullptr, nullptr, 0, wave_prio, nullptr, blk, n_elem_blocks, 0 WGS_ARG);
    // ---- the last workgroup to finish hands S | b to the host
    __threadfence();
    __syncthreads();
    if (tid == 0) s_flag = atomicAdd(&ctl->ticket, 1u) == (unsigned)(it + 1) * (unsigned)nblk - 1u ? 1 : 0;
    __syncthreads();
    if (s_flag) {
      __threadfence();
      for (int k = tid; k < NE; k += 256) result_host[k] = S[k];  // S | b are contiguous
      __threadfence_system();
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(result_host + NE), want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ------------------------------------------------------------------------------------------ blocked dense solve
// Systems with more than 64 unknowns (the pose graph: 3 unknowns per keyframe): right-looking BLOCKED elimination over
// the whole device.  Every matrix element still receives the reference's updates a_ij -= f_i(k) * r_kj(k) one pivot at
// a time, in ascending k, with the same operands (separate multiply and subtract, no FMA) -- only the schedule changes:
//   k_lu_panel   one workgroup factors LU_NB columns: pivot search (reference's first-maximum rule), row swaps inside the
//                panel, division of the pivot row, update of the panel columns; the multipliers f_i(k) = a_ik stay in
//                place of the entries the reference would overwrite with values it never reads again;
//   k_lu_urow    one thread per trailing column (b rides along as column n): the panel's row swaps, then the LU_NB
//                pivot rows of the block, each normalised by its saved pivot and applied to the block rows below it;
//   k_lu_update  all trailing rows x columns: the LU_NB delayed updates of each element, in order, from LDS tiles;
//   k_lu_backsub one workgroup: the reference's ascending-j subtraction chain per row (strictly serial by its order),
//                fed with products A(i,j)*x[j] that all rows form as soon as x[j] exists.
// The |f| < 1e-18 skip (dense.hpp:80) is re-evaluated from the stored multiplier wherever it is applied.
#define LU_NB 32
__global__ void k_lu_init(const double* __restrict__ A, const double* __restrict__ b, int n, double* __restrict__ Wm, int* __restrict__ status) {
  const int ld = n + 1;
  const size_t total = (size_t)n * ld;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / ld), j = (int)(e % ld);
    Wm[e] = j < n ? A[(size_t)i * n + j] : b[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 0;
}

// The panel is worked on in a COLUMN-major copy Pn[LU_NB][m] (m = n - k0 rows): the pivot search and the row-parallel
// updates then touch consecutive addresses.  IN_LDS: the copy lives in LDS (m * LU_NB doubles <= 128 KiB), otherwise in
// a global scratch area.
#define LU_PANEL_THREADS 256
template <bool IN_LDS>
__global__ __launch_bounds__(LU_PANEL_THREADS) void k_lu_panel(double* __restrict__ Wm, int n, int k0, int* __restrict__ piv_out,
                                                               double* __restrict__ akk_out, int* __restrict__ status,
                                                               double* __restrict__ gscratch) {
  extern __shared__ __align__(16) double lds_panel[];
  __shared__ double rrow[LU_NB];
  __shared__ double red_v[2][LU_PANEL_THREADS / 64];
  __shared__ int red_i[2][LU_PANEL_THREADS / 64];
  if (status[0]) return;
  const int ld = n + 1, tid = threadIdx.x, nt = blockDim.x;
  const int k1 = min(k0 + LU_NB, n), nbr = k1 - k0, m = n - k0;
  double* Pn;
  if constexpr (IN_LDS) Pn = lds_panel; else Pn = gscratch;
  for (int e = tid; e < m * nbr; e += nt) {  // e = r * nbr + c: each row's panel entries are contiguous in Wm
    const int r = e / nbr, c = e % nbr;
    Pn[(size_t)c * m + r] = Wm[(size_t)(k0 + r) * ld + k0 + c];
  }
  __syncthreads();
  // pivot candidates of step 0; later steps get theirs from the update loop of the step before
  double bv = -1.0;
  int bi = 0x7fffffff;
  for (int r = tid; r < m; r += nt) {
    const double v = fabs(Pn[r]);
    if (v > bv) { bv = v; bi = r; }  // NaN never wins `v > best`
  }
  for (int kk = 0; kk < nbr; kk++) {
    const int k = k0 + kk;
    double* colk = Pn + (size_t)kk * m;
    // ---- pivot (dense.hpp:61-67): first maximum of |a_ik| over rows i >= k; every thread finishes the reduction itself
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_down(bv, o, 64);
      const int oi = __shfl_down(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int par = kk & 1;  // double-buffered so that one barrier per reduction suffices
    if ((tid & 63) == 0) { red_v[par][tid >> 6] = bv; red_i[par][tid >> 6] = bi; }
    __syncthreads();
    double v = red_v[par][0];
    int piv = red_i[par][0];
#pragma unroll
    for (int w = 1; w < LU_PANEL_THREADS / 64; w++)
      if (red_v[par][w] > v || (red_v[par][w] == v && red_i[par][w] < piv)) { v = red_v[par][w]; piv = red_i[par][w]; }
    const double akk0 = fabs(colk[kk]);
    double best = v;
    if (akk0 != akk0) { piv = kk; best = akk0; }  // a NaN on the diagonal stays the pivot
    else if (piv == 0x7fffffff) { piv = kk; best = akk0; }
    if (best < 1e-15) {  // uniform: every thread evaluated the same data
      if (tid == 0) status[0] = 1;
      return;
    }
    const double akk = colk[piv];  // A(k,k) after the swap (dense.hpp:74)
    if (tid == 0) { piv_out[k] = k0 + piv; akk_out[k] = akk; }
    __syncthreads();  // everybody has read column k before the swap rewrites it
    // ---- row swap inside the panel (all its columns: the multipliers of earlier panel steps travel with their row)
    // fused with the normalisation of the pivot row (dense.hpp:69-75): thread c owns column c
    if (tid < nbr) {
      double* cj = Pn + (size_t)tid * m;
      const double top = cj[kk], low = cj[piv];
      const double newk = tid >= kk ? low / akk : low;
      cj[kk] = newk;
      if (piv != kk) cj[piv] = top;
      rrow[tid] = newk;
    }
    __syncthreads();
    // ---- update the panel columns right of k (dense.hpp:78-82); column k keeps the multiplier.  The next step's
    // pivot candidates (column kk+1, rows > kk) are taken from the values just written.
    bv = -1.0;
    bi = 0x7fffffff;
    double* cnext = Pn + (size_t)(kk + 1) * m;
    for (int r = kk + 1 + tid; r < m; r += nt) {
      const double f = colk[r];
      if (!(fabs(f) < 1e-18)) {
        for (int c = kk + 1; c < nbr; c++) {
          double* cc = Pn + (size_t)c * m;
          cc[r] = cc[r] - f * rrow[c];
        }
      }
      if (kk + 1 < nbr) {
        const double vv = fabs(cnext[r]);
        if (vv > bv) { bv = vv; bi = r; }
      }
    }
    // (no barrier here: the next reduction's barrier orders these writes before any cross-thread read)
  }
  __syncthreads();
  for (int e = tid; e < m * nbr; e += nt) {
    const int r = e / nbr, c = e % nbr;
    Wm[(size_t)(k0 + r) * ld + k0 + c] = Pn[(size_t)c * m + r];
  }
}

__global__ __launch_bounds__(256) void k_lu_urow(double* __restrict__ Wm, int n, int k0, const int* __restrict__ piv, const double* __restrict__ akk,
                                                 const int* __restrict__ status) {
  __shared__ double Lb[LU_NB][LU_NB + 1];  // multipliers of the block rows: Lb[ii][kk] = a(k0+ii, k0+kk), kk < ii
  __shared__ double pv[LU_NB];
  __shared__ int src_blk[LU_NB];           // row whose content ends up in block row kk after the panel's swaps
  __shared__ int out_row[LU_NB], out_src[LU_NB], n_out;  // rows below the block that receive other content
  __shared__ int pos[2 * LU_NB], cur[2 * LU_NB];
  if (status[0]) return;
  const int ld = n + 1, k1 = min(k0 + LU_NB, n), nbr = k1 - k0;
  for (int e = threadIdx.x; e < LU_NB * LU_NB; e += blockDim.x) {
    const int ii = e / LU_NB, kk = e % LU_NB;
    Lb[ii][kk] = (ii < nbr && kk < nbr) ? Wm[(size_t)(k0 + ii) * ld + k0 + kk] : 0.0;
  }
  if (threadIdx.x < LU_NB) pv[threadIdx.x] = threadIdx.x < nbr ? akk[k0 + threadIdx.x] : 1.0;
  if (threadIdx.x == 0) {
    // the swaps (dense.hpp:69-72) as one permutation: cur[q] = original row now sitting at tracked position q
    int np_ = 0;  // pos / cur live in LDS: dynamically indexed private arrays would go to scratch memory
    for (int kk = 0; kk < nbr; kk++) { pos[np_] = k0 + kk; cur[np_] = k0 + kk; np_++; }
    for (int kk = 0; kk < nbr; kk++) {
      const int p = piv[k0 + kk];
      if (p == k0 + kk) continue;
      int q = -1;
      for (int s = 0; s < np_; s++)
        if (pos[s] == p) q = s;
      if (q < 0) { q = np_; pos[np_] = p; cur[np_] = p; np_++; }
      const int tmp = cur[kk];
      cur[kk] = cur[q];
      cur[q] = tmp;
    }
    int no = 0;
    for (int s = 0; s < np_; s++) {
      if (s < nbr) src_blk[s] = cur[s];
      else if (cur[s] != pos[s]) { out_row[no] = pos[s]; out_src[no] = cur[s]; no++; }
    }
    n_out = no;
  }
  __syncthreads();
  const int j = k1 + blockIdx.x * blockDim.x + threadIdx.x;  // trailing column, n = the right-hand side
  if (j > n) return;
  // all loads first (independent), then the stores: sources and destinations are the same set of rows
  double col[LU_NB], spill[LU_NB];
  const int no = n_out;
#pragma unroll
  for (int kk = 0; kk < LU_NB; kk++) col[kk] = kk < nbr ? Wm[(size_t)src_blk[kk] * ld + j] : 0.0;
#pragma unroll
  for (int s = 0; s < LU_NB; s++) spill[s] = s < no ? Wm[(size_t)out_src[s] * ld + j] : 0.0;
#pragma unroll
  for (int s = 0; s < LU_NB; s++)
    if (s < no) Wm[(size_t)out_row[s] * ld + j] = spill[s];
#pragma unroll
  for (int kk = 0; kk < LU_NB; kk++) {
    if (kk < nbr) {
      const double r = col[kk] / pv[kk];  // dense.hpp:75-76
      col[kk] = r;
#pragma unroll
      for (int ii = kk + 1; ii < LU_NB; ii++) {
        const double f = Lb[ii][kk];
        if (ii < nbr && !(fabs(f) < 1e-18)) col[ii] = col[ii] - f * r;  // dense.hpp:79-83
      }
    }
  }
#pragma unroll
  for (int kk = 0; kk < LU_NB; kk++)
    if (kk < nbr) Wm[(size_t)(k0 + kk) * ld + j] = col[kk];
}

#define LU_TILE 16
__global__ __launch_bounds__(LU_TILE* LU_TILE) void k_lu_update(double* __restrict__ Wm, int n, int k0, const int* __restrict__ status) {
  __shared__ double Lt[LU_TILE][LU_NB + 1];  // multipliers of this tile's rows
  __shared__ double Ut[LU_NB][LU_TILE + 1];  // normalised pivot rows of this tile's columns
  if (status[0]) return;
  const int ld = n + 1, k1 = min(k0 + LU_NB, n), nbr = k1 - k0;
  const int tx = threadIdx.x % LU_TILE, ty = threadIdx.x / LU_TILE;
  const int i0 = k1 + blockIdx.y * LU_TILE, j0 = k1 + blockIdx.x * LU_TILE;
  for (int e = threadIdx.x; e < LU_TILE * LU_NB; e += LU_TILE * LU_TILE) {
    const int r = e / LU_NB, kk = e % LU_NB;
    Lt[r][kk] = (i0 + r < n && kk < nbr) ? Wm[(size_t)(i0 + r) * ld + k0 + kk] : 0.0;
  }
  for (int e = threadIdx.x; e < LU_NB * LU_TILE; e += LU_TILE * LU_TILE) {
    const int kk = e / LU_TILE, c = e % LU_TILE;
    Ut[kk][c] = (j0 + c <= n && kk < nbr) ? Wm[(size_t)(k0 + kk) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  const int i = i0 + ty, j = j0 + tx;
  if (i >= n || j > n) return;
  double acc = Wm[(size_t)i * ld + j];
#pragma unroll 8
  for (int kk = 0; kk < nbr; kk++) {
    const double f = Lt[ty][kk];
    if (!(fabs(f) < 1e-18)) acc = acc - f * Ut[kk][tx];  // dense.hpp:79-83, pivots k0+kk in ascending order
  }
  Wm[(size_t)i * ld + j] = acc;
}

// back-substitution (dense.hpp:86-91) on the eliminated system; unit diagonal.  The subtraction chain of row i is
// strictly serial and can only start when x[i+1] exists, so everything else is taken off that path: while thread 0
// runs the chain of row i out of LDS, the other threads stage row i-1 -- the products A(i-1,j)*x[j] for the columns
// whose x is known (x lives in LDS), the raw A(i-1,i) for the one that is not; thread 0 forms that last product itself.
// IN_LDS = false (n > SOLVE_LDS_MAX_N: x and the two staged rows no longer fit in 156 KiB): the same schedule with the
// three arrays in a global scratch buffer (same workgroup, so __syncthreads() orders the hand-over); slower per
// element, still the reference's subtraction chain.
template <bool IN_LDS>
__global__ __launch_bounds__(512) void k_lu_backsub(const double* __restrict__ Wm, int n, double* __restrict__ x, int* __restrict__ status,
                                                    double* __restrict__ gscratch) {
  extern __shared__ __align__(16) double bs_dyn[];  // x[n] | row buffers [2][n]
  if (status[0]) return;
  double* bs_lds = IN_LDS ? bs_dyn : gscratch;
  double* xs = bs_lds;
  const int ld = n + 1, tid = threadIdx.x, nt = blockDim.x;
  for (int i = n - 1; i >= 0; i--) {
    double* cur = bs_lds + (size_t)n * (1 + (i & 1));        // row i: staged one iteration ago
    double* nxt = bs_lds + (size_t)n * (1 + ((i + 1) & 1));  // row i-1
    if (tid == 0) {
      double s = Wm[(size_t)i * ld + n];
      if (i + 1 < n) {
        s -= cur[i + 1] * xs[i + 1];  // the newest product (dense.hpp:89)
        int c = i + 2;
        for (; c + 16 <= n; c += 16) {
          double p[16];
#pragma unroll
          for (int q = 0; q < 16; q++) p[q] = cur[c + q];
#pragma unroll
          for (int q = 0; q < 16; q++) s -= p[q];
        }
        for (; c < n; c++) s -= cur[c];
      }
      xs[i] = s;
      x[i] = s;
    } else if (i >= 1) {
      const double* rowm = Wm + (size_t)(i - 1) * ld;
      for (int c = i + (tid - 1); c < n; c += nt - 1) {
        const double v = rowm[c];
        nxt[c] = (c == i) ? v : v * xs[c];  // xs[c], c > i, was written in an earlier iteration
      }
    }
    __syncthreads();
  }
}

#define SOLVE_LDS_MAX_N 6400  // back-substitution keeps x and two staged rows in LDS while 3 * n * 8 <= 156 KiB

// n == 36 / 60: rows in registers.  n <= 64: wave-synchronous LDS kernel.  Larger systems (pose graphs): blocked
// elimination over the whole device on a working copy [n][n+1] in ctx->d[7].
static int launch_solve_kernels(sfmx_ctx* c, const double* dA, const double* db, int n, double* dx, int* dstatus);
static int launch_solve(sfmx_ctx* c, const double* dA, const double* db, int n, double* dx, int* dstatus) {
  prof_begin(c, KID_SOLVE);
  const int rc = launch_solve_kernels(c, dA, db, n, dx, dstatus);
  prof_end(c);
  return rc;
}
static int launch_solve_kernels(sfmx_ctx* c, const double* dA, const double* db, int n, double* dx, int* dstatus) {
  static const bool stamps = getenv("SFMX_SOLVE_STAMPS") != nullptr;
  const bool rows16 = ((reinterpret_cast<uintptr_t>(dA) & 15) == 0);  // k_solve_regs fetches rows with 16-byte loads
  if (n == 36 && stamps && rows16) {  // diagnostic: per-phase cycle stamps of one solve on stderr
    unsigned long long* d = nullptr;
    unsigned long long h[40 + 3 * 36] = {};
    if (hipMalloc(&d, sizeof(h)) == hipSuccess) {
      (void)hipMemsetAsync(d, 0, sizeof(h), c->stream);
      k_solve_regs_stamps<36><<<1, 64, 0, c->stream>>>(dA, db, dx, dstatus, d);
      (void)hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
      (void)hipStreamSynchronize(c->stream);
      (void)hipFree(d);
      fprintf(stderr, "solve36 stamps: load %llu |", h[1] - h[0]);
      for (int k = 0; k < 36; k++) fprintf(stderr, " %llu", h[3 + k] - h[2 + k]);
      fprintf(stderr, " | backsub %llu | total %llu\n", h[39] - h[38], h[39] - h[0]);
      for (int k = 0; k < 36; k += 7)
        fprintf(stderr, "  step %2d: pivot %llu | row write %llu | division %llu | eliminate %llu\n", k, h[40 + 3 * k] - h[2 + k],
                h[41 + 3 * k] - h[40 + 3 * k], h[42 + 3 * k] - h[41 + 3 * k], h[3 + k] - h[42 + 3 * k]);
    }
  } else if (n == 36 && rows16) {  // window of 6 (the reference default) and of 10 (C4): rows in registers
    k_solve_regs<36><<<1, 64, 0, c->stream>>>(dA, db, dx, dstatus);
  } else if (n == 60 && rows16) {
    k_solve_regs<60><<<1, 64, 0, c->stream>>>(dA, db, dx, dstatus);
  } else if (n <= SOLVE_WAVE_MAX_N) {
    k_solve_wave<<<1, 64, 0, c->stream>>>(dA, db, n, dx, dstatus);
  } else {
    const size_t wbytes = (size_t)n * (n + 1) * 8, abytes = (size_t)n * 8, pbytes = (((size_t)n * 4) + 15) & ~(size_t)15;
    const size_t sbytes = (size_t)n * LU_NB * 8;  // column-major panel copy when it does not fit in LDS
    const int lds_max_n = getenv("SFMX_BACKSUB_LDS_MAX_N") ? atoi(getenv("SFMX_BACKSUB_LDS_MAX_N")) : SOLVE_LDS_MAX_N;  // test hook
    const bool bs_in_lds = n <= lds_max_n && n <= SOLVE_LDS_MAX_N;
    const size_t gbytes = bs_in_lds ? 0 : (size_t)3 * n * 8;
    SFMX_HIP(c, c->d[7].ensure(wbytes + abytes + pbytes + sbytes + gbytes + 64));
    double* Wm = c->d[7].as<double>();
    double* akk = reinterpret_cast<double*>(c->d[7].as<char>() + wbytes);
    int* piv = reinterpret_cast<int*>(c->d[7].as<char>() + wbytes + abytes);
    double* scratch = reinterpret_cast<double*>(c->d[7].as<char>() + wbytes + abytes + pbytes);
    constexpr size_t kPanelLds = 156 * 1024;  // 160 KiB of LDS per workgroup minus the kernel's static arrays
    {  // the dynamic-LDS opt-in is a per-device function attribute; contexts of several host threads may get here at once
      static std::mutex attr_mu;
      static bool attr_set[64] = {};
      std::lock_guard<std::mutex> lk(attr_mu);
      const int dev = c->device & 63;
      if (!attr_set[dev]) {
        SFMX_HIP(c, hipFuncSetAttribute((const void*)k_lu_panel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelLds));
        SFMX_HIP(c, hipFuncSetAttribute((const void*)k_lu_backsub<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelLds));
        attr_set[dev] = true;
      }
    }
    k_lu_init<<<256, 256, 0, c->stream>>>(dA, db, n, Wm, dstatus);
    for (int k0 = 0; k0 < n; k0 += LU_NB) {
      const int k1 = k0 + LU_NB < n ? k0 + LU_NB : n;
      const size_t pan = (size_t)(n - k0) * LU_NB * 8;
      if (pan <= kPanelLds) k_lu_panel<true><<<1, LU_PANEL_THREADS, pan, c->stream>>>(Wm, n, k0, piv, akk, dstatus, nullptr);
      else k_lu_panel<false><<<1, LU_PANEL_THREADS, 0, c->stream>>>(Wm, n, k0, piv, akk, dstatus, scratch);
      const int tcols = n - k1 + 1;  // trailing columns incl. the right-hand side
      k_lu_urow<<<(tcols + 255) / 256, 256, 0, c->stream>>>(Wm, n, k0, piv, akk, dstatus);
      if (k1 < n) {
        const dim3 grid((tcols + LU_TILE - 1) / LU_TILE, (n - k1 + LU_TILE - 1) / LU_TILE);
        k_lu_update<<<grid, LU_TILE * LU_TILE, 0, c->stream>>>(Wm, n, k0, dstatus);
      }
    }
    if (bs_in_lds) k_lu_backsub<true><<<1, 512, (size_t)3 * n * 8, c->stream>>>(Wm, n, dx, dstatus, nullptr);
    else k_lu_backsub<false><<<1, 512, 0, c->stream>>>(Wm, n, dx, dstatus,
                                                       reinterpret_cast<double*>(c->d[7].as<char>() + wbytes + abytes + pbytes + sbytes));
  }
  SFMX_HIP(c, hipGetLastError());
  return SFMX_OK;
}

// T:1064-1071 on a reduced system that was summed without it: S_ii += lambda, then the gauge on DoF 0..5
__global__ void k_ba_damp_gauge(double* __restrict__ S, double* __restrict__ b, int D, double lambda) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D) return;
  double v = S[(size_t)i * D + i] + lambda;
  if (i < 6) { v = v + 1e9; b[i] = 0.0; }
  S[(size_t)i * D + i] = v;
}

// dx | status of one BA step straight into pinned host memory, then a sequence word (system-scope release): the host
// polls that word instead of going through a DMA copy and a stream synchronisation (~25 us per BA iteration)
__global__ void k_ba_publish(const double* __restrict__ work, int D, double* __restrict__ host_out, unsigned long long seq) {
  for (int i = threadIdx.x; i <= D; i += blockDim.x) host_out[i] = work[i];  // [D] holds the status word
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_out + D + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// fused: 0 = sums only; otherwise the last workgroup of the reduction also solves and publishes (D = 36 / 60 only: see
// ba_can_fuse_solve), to q->work and, if host_out is given, to pinned host memory with sequence word `seq`
static bool ba_can_fuse_solve(const sfmx_ba_problem* q) { return q->W == 6 || q->W == 10; }
// Shape of a problem: window-sized ones (or SFMX_BA_EXPAND=merged) get their contribution rows from the points kernel itself, all
// rows resident; large ones (C4: 50 000 points x 32 KB) expand CHUNKS of points into a two-slot ring that stays inside the
// 256 MB Infinity Cache while the reduction consumes the other slot (ba_launch_reduce).
static bool ba_merged(const sfmx_ba_problem* q) {
  static const char* expand_env = getenv("SFMX_BA_EXPAND");  // "split" / "merged": A/B and tests
  return expand_env ? expand_env[0] == 'm' : q->P <= BA_MERGED_EXPAND_MAX_P;
}
static int ba_chunk_points() {
  static const int v = getenv("SFMX_BA_CHUNK") ? atoi(getenv("SFMX_BA_CHUNK")) : 4096;
  return v < 128 ? 128 : v;
}
// points phase of one iteration: the records of every point (and, in the merged shape, their contribution rows)
static int ba_launch_points(sfmx_ctx* c, sfmx_ba_problem* q, const double* d_poses, double fx, double fy, double cx, double cy, double huber,
                            int wave_prio) {
  // SFMX_BA_POINTS=global: the window kernel with its records in global memory (A/B and tests; identical rows)
  const char* pts_env = getenv("SFMX_BA_POINTS");
  // points per workgroup of the LDS kernel (SFMX_BA_PTS=1|2|4): the record phases keep 16 lanes per point busy whatever the count,
  // the row expansion is the workgroup's throughput-bound part and scales with it
  int pts_wg = 2;
  if (const char* e = getenv("SFMX_BA_PTS")) pts_wg = atoi(e);
  if (pts_wg != 1 && pts_wg != 4) pts_wg = 2;
  const size_t rec_lds = (size_t)pts_wg * q->MS * BA_SLOT * 8 + 16;
  if (ba_merged(q) && q->lds_points && rec_lds <= 40960 && !(pts_env && pts_env[0] == 'g')) {
    const int nwg = (q->P + pts_wg - 1) / pts_wg;
#define BA_PL_ARGS q->W, q->P, q->MS, d_poses, q->X, q->obs_ptr, q->obs_li, q->obs_uv, fx, fy, cx, cy, huber, q->rec, q->slot_of, q->contrib, wave_prio
#define BA_PL_LAUNCH(WT_)                                                                                                     \
  do {                                                                                                                        \
    if (pts_wg == 1) SFMX_PROF(c, KID_BA_POINTS, (k_ba_points_window_lds<WT_, 1><<<nwg, 256, rec_lds, c->stream>>>(BA_PL_ARGS)));      \
    else if (pts_wg == 2) SFMX_PROF(c, KID_BA_POINTS, (k_ba_points_window_lds<WT_, 2><<<nwg, 256, rec_lds, c->stream>>>(BA_PL_ARGS))); \
    else SFMX_PROF(c, KID_BA_POINTS, (k_ba_points_window_lds<WT_, 4><<<nwg, 256, rec_lds, c->stream>>>(BA_PL_ARGS)));                  \
  } while (0)
    if (q->W == 6) BA_PL_LAUNCH(6);
    else BA_PL_LAUNCH(0);
#undef BA_PL_LAUNCH
#undef BA_PL_ARGS
  } else if (ba_merged(q)) {
    SFMX_PROF(c, KID_BA_POINTS, (k_ba_points_window<<<(q->P + BA_PTS - 1) / BA_PTS, 256, 0, c->stream>>>(
                                    q->W, q->P, q->MS, d_poses, q->X, q->obs_ptr, q->obs_li, q->obs_uv, fx, fy, cx, cy, huber, q->rec, q->slot_of, q->contrib, wave_prio)));
  } else {
    SFMX_PROF(c, KID_BA_POINTS, (k_ba_points_bulk<<<(q->P + 63) / 64, 64, 0, c->stream>>>(q->W, q->P, q->MS, d_poses, q->X, q->obs_ptr, q->obs_li, q->obs_uv, fx,
                                                                                         fy, cx, cy, huber, q->rec, q->slot_of, nullptr, wave_prio)));
  }
  return SFMX_OK;
}
static int ba_element_blocks(const sfmx_ba_problem* q) { const int D = 6 * q->W; return (D * D + D + BAR_COLS - 1) / BAR_COLS; }
// one reduction launch over `p_cnt` rows starting at `rows`
static void ba_reduce_kernel(sfmx_ctx* c, sfmx_ba_problem* q, const double* rows, int p_cnt, double lambda, int damp, double* S_out, double* b_out,
                             const double* init, bool fused_solve, double* host_out, unsigned long long seq, int wave_prio, int wg_lo, int nwg,
                             int publish_system, bool streaming) {
  // Rows per tile of the window shape (SFMX_BA_TILE=64|32|16; the register ring always holds 128 rows of look-ahead): 33 / 16 / 11 KB
  // of LDS per workgroup.  A workgroup of the reduction starts when a CU has that much LDS free next to the resident KLT waves
  // (22.6 KB each, up to seven per CU): with the BA chain bounding a pass again, 32-row tiles are 1.5 % faster than 64-row ones
  // (profiles/r03_ab_inproc_lds.txt); identical sums.
  const char* tile_env = getenv("SFMX_BA_TILE");
  const int tile_rows = tile_env ? atoi(tile_env) : 32;
  if (fused_solve && q->W == 6 && tile_rows == 32) {
    k_ba_reduce<32, 36, 4, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, q->ticket, q->work, host_out, seq, wave_prio, init,
                                                         wg_lo, publish_system);
  } else if (fused_solve && q->W == 6 && tile_rows == 16) {
    k_ba_reduce<16, 36, 8, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, q->ticket, q->work, host_out, seq, wave_prio, init,
                                                         wg_lo, publish_system);
  } else if (fused_solve && q->W == 6) {
    k_ba_reduce<64, 36, 2, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, q->ticket, q->work, host_out, seq, wave_prio, init,
                                                         wg_lo, publish_system);
  } else if (fused_solve && q->W == 10) {
    k_ba_reduce<128, 60, 2, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, q->ticket, q->work, host_out, seq, wave_prio, init,
                                                          wg_lo, publish_system);
  } else if (!streaming) {
    k_ba_reduce<64, 0, 2, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, nullptr, nullptr, nullptr, 0, wave_prio, init, wg_lo);
  } else {
    k_ba_reduce<128, 0, 2, 2><<<nwg, 256, 0, c->stream>>>(q->W, p_cnt, rows, lambda, damp, S_out, b_out, nullptr, nullptr, nullptr, 0, wave_prio, init, wg_lo);
  }
}
// Reduction phase over the points [p_lo, p_lo + p_cnt) into S_out | b_out; init: see k_ba_reduce.
// wg_lo / wg_cnt, e_lo / e_hi (element-sharded step): only the element blocks [wg_lo, wg_lo + wg_cnt) of BAR_COLS elements are
// reduced, and only the row entries that feed elements [e_lo, e_hi) are expanded.
// Large problems: the rows of a chunk of points are expanded on the context's second stream into one slot of a two-slot ring
// while the reduction kernel of the chunk before it runs on the first; every element's chain continues from chunk to chunk in
// S_out itself (init = S_out), so the sums are the reference's sequence over all points and the 1.6 GB that a C4 iteration used to
// write and read back never leave the Infinity Cache (2 x 134 MB at W = 10).
static int ba_launch_reduce(sfmx_ctx* c, sfmx_ba_problem* q, int p_lo, int p_cnt, double lambda, int damp, double* S_out, double* b_out,
                            const double* init, bool fused_solve, double* host_out, unsigned long long seq, int wave_prio, int wg_lo = 0,
                            int wg_cnt = -1, int publish_system = 0, int e_lo = 0, int e_hi = 0x7fffffff) {
  const size_t CS = (size_t)36 * q->W * q->W + 48 * q->W;
  const int nwg = wg_cnt >= 0 ? wg_cnt : ba_element_blocks(q);
  if (nwg == 0 || p_cnt <= 0) return SFMX_OK;
  if (wg_cnt >= 0) fused_solve = false;  // a slice of the system: nothing to solve yet
  if (q->chunk == 0) {  // all rows resident (written by k_ba_points_window)
    SFMX_PROF(c, KID_BA_REDUCE, ba_reduce_kernel(c, q, q->contrib + (size_t)p_lo * CS, p_cnt, lambda, damp, S_out, b_out, init, fused_solve, host_out, seq,
                                                 wave_prio, wg_lo, nwg, publish_system, q->P > BA_MERGED_EXPAND_MAX_P));
    return SFMX_OK;
  }
  if (!q->ev_sync) {
    SFMX_HIP(c, hipEventCreateWithFlags(&q->ev_sync, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
      SFMX_HIP(c, hipEventCreateWithFlags(&q->ev_exp[k], hipEventDisableTiming));
      SFMX_HIP(c, hipEventCreateWithFlags(&q->ev_red[k], hipEventDisableTiming));
    }
  }
  prof_begin(c, KID_BA_REDUCE);  // expansion (second stream) + reduction of all chunks as one profile entry
  // the second stream starts behind everything queued so far: the points kernel, and earlier reductions that read the ring
  SFMX_HIP(c, hipEventRecord(q->ev_sync, c->stream));
  SFMX_HIP(c, hipStreamWaitEvent(c->copy_stream, q->ev_sync, 0));
  const int chunk = q->chunk, p_end = p_lo + p_cnt;
  int k = 0;
  for (int p0 = p_lo; p0 < p_end; p0 += chunk, ++k) {
    const int cnt = p_end - p0 < chunk ? p_end - p0 : chunk;
    const bool last = p0 + cnt >= p_end;
    double* slot = q->contrib + (size_t)(k & 1) * chunk * CS;
    if (k >= 2) SFMX_HIP(c, hipStreamWaitEvent(c->copy_stream, q->ev_red[k & 1], 0));  // the reduction that read this slot is done
    k_ba_expand<<<cnt, 256, 0, c->copy_stream>>>(q->W, q->P, q->MS, q->rec, q->slot_of, slot, e_lo, e_hi, p0);
    SFMX_HIP(c, hipEventRecord(q->ev_exp[k & 1], c->copy_stream));
    SFMX_HIP(c, hipStreamWaitEvent(c->stream, q->ev_exp[k & 1], 0));
    ba_reduce_kernel(c, q, slot, cnt, lambda, last ? damp : 0, S_out, b_out, k == 0 ? init : S_out, last && fused_solve, host_out, seq, wave_prio, wg_lo,
                     nwg, publish_system, true);
    SFMX_HIP(c, hipEventRecord(q->ev_red[k & 1], c->stream));
  }
  prof_end(c);
  return SFMX_OK;
}
static const double* ba_stage_poses(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, bool zero_copy_poses, int* rc_out) {
  *rc_out = SFMX_OK;
  if (c->h[0].ensure((size_t)q->W * 96) != hipSuccess) { *rc_out = sfmx_fail(c, SFMX_ERR_HIP, "pinned pose staging", hipErrorOutOfMemory); return nullptr; }
  memcpy(c->h[0].p, poses_wc, (size_t)q->W * 96);
  // zero_copy_poses: k_ba_points reads the 96 W bytes straight out of the pinned staging buffer (the caller does not touch
  // it again before it has seen this step's result); otherwise one DMA copy into HBM first
  if (zero_copy_poses) return c->h[0].as<double>();
  const hipError_t e = hipMemcpyAsync(q->poses, c->h[0].p, (size_t)q->W * 96, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) { *rc_out = sfmx_fail(c, SFMX_ERR_HIP, "hipMemcpyAsync(poses)", e); return nullptr; }
  return q->poses;
}
static int ba_wave_prio() {
  static const int wave_prio = getenv("SFMX_BA_NO_WAVE_PRIO") ? 0 : 1;
  return wave_prio;
}
static int ba_launch_build(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx, double cy,
                           double huber, double lambda, int damp, KernelTimer& t, bool zero_copy_poses = false, bool fused_solve = false,
                           double* host_out = nullptr, unsigned long long seq = 0, int publish_system = 0) {
  int rc = SFMX_OK;
  const double* d_poses = ba_stage_poses(c, q, poses_wc, zero_copy_poses, &rc);
  if (rc) return rc;
  t.start();
  rc = ba_launch_points(c, q, d_poses, fx, fy, cx, cy, huber, ba_wave_prio());
  if (rc) return rc;
  rc = ba_launch_reduce(c, q, 0, q->P, lambda, damp, q->S, q->b, nullptr, fused_solve, host_out, seq, ba_wave_prio(), 0, -1, publish_system);
  if (rc) return rc;
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  return SFMX_OK;
}

// ---- virtual world (TEST MODE, SFMX_VIRTUAL_WORLD=N): what N ranks of the point-sharded step compute, on one GPU ----------
// The N contiguous point ranges sfmx_shard_range hands out are reduced separately (each chain starts at +0.0, as on its own
// rank) and combined per element in the order SFMX_VIRTUAL_WORLD_ORDER names -- the association an all-reduce may use:
//   rank (default)  ((p0 + p1) + p2) + ...          reverse  ((pN-1 + pN-2) + ...) + p0
//   ring            chunk k of the buffer starts at rank k+1 (ring reduce-scatter)      tree  pairwise ((p0+p1) + (p2+p3)) + ...
//   relay           no partials at all: shard r continues the running sums of shard r-1 (the sequential reference order)
__global__ void k_ba_combine_partials(const double* __restrict__ part, int n_ranks, int ne, int order, double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  auto P = [&](int r) { return part[(size_t)r * ne + e]; };
  double acc;
  if (order == 1) {
    acc = P(n_ranks - 1);
    for (int r = n_ranks - 2; r >= 0; --r) acc = acc + P(r);
  } else if (order == 2) {
    const int chunk = (ne + n_ranks - 1) / n_ranks, k = e / chunk;
    acc = P((k + 1) % n_ranks);
    for (int s = 2; s <= n_ranks; ++s) acc = acc + P((k + s) % n_ranks);
  } else if (order == 3) {
    double v[64];
    int m = n_ranks < 64 ? n_ranks : 64;
    for (int r = 0; r < m; ++r) v[r] = P(r);
    while (m > 1) {
      int o = 0;
      for (int r = 0; r + 1 < m; r += 2) v[o++] = v[r] + v[r + 1];
      if (m & 1) v[o++] = v[m - 1];
      m = o;
    }
    acc = v[0];
  } else {
    acc = P(0);
    for (int r = 1; r < n_ranks; ++r) acc = acc + P(r);
  }
  out[e] = acc;
}
static int ba_virtual_world() {
  const char* e = getenv("SFMX_VIRTUAL_WORLD");  // read per call: tests switch it inside one process
  const int n = e ? atoi(e) : 0;
  return n > 1 && n <= 64 ? n : 0;
}
static int ba_build_virtual_world(sfmx_ctx* c, sfmx_ba_problem* q, int n_ranks, const double* poses_wc, double fx, double fy, double cx,
                                  double cy, double huber, KernelTimer& t) {
  const char* oe = getenv("SFMX_VIRTUAL_WORLD_ORDER");
  const std::string order_s = oe ? oe : "rank";
  const int order = order_s == "reverse" ? 1 : order_s == "ring" ? 2 : order_s == "tree" ? 3 : order_s == "relay" ? 4 : 0;
  const int D = 6 * q->W, NE = D * D + D;
  int rc = SFMX_OK;
  const double* d_poses = ba_stage_poses(c, q, poses_wc, false, &rc);
  if (rc) return rc;
  SFMX_HIP(c, q->bufs[11].ensure((size_t)n_ranks * NE * 8));
  double* part = q->bufs[11].as<double>();
  t.start();
  rc = ba_launch_points(c, q, d_poses, fx, fy, cx, cy, huber, ba_wave_prio());
  if (rc) return rc;
  for (int r = 0; r < n_ranks; ++r) {
    int lo = 0, hi = 0;
    sfmx_shard_range(q->P, r, n_ranks, &lo, &hi);
    if (order == 4) rc = ba_launch_reduce(c, q, lo, hi - lo, 0.0, 0, q->S, q->b, r == 0 ? nullptr : q->S, false, nullptr, 0, ba_wave_prio());
    else rc = ba_launch_reduce(c, q, lo, hi - lo, 0.0, 0, part + (size_t)r * NE, part + (size_t)r * NE + (size_t)D * D, nullptr, false, nullptr, 0,
                               ba_wave_prio());
    if (rc) return rc;
  }
  if (order != 4) k_ba_combine_partials<<<(NE + 255) / 256, 256, 0, c->stream>>>(part, n_ranks, NE, order, q->S);
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  return SFMX_OK;
}

extern "C" {

int sfmx_ba_reset(sfmx_ctx* c, sfmx_ba_problem* q, int W, int P, const double* X, const int32_t* obs_ptr, const int32_t* obs_li,
                  const double* obs_uv) {
  SFMX_REQUIRE(c, c && q && W >= 1 && W <= BA_MAX_W && P >= 1 && X && obs_ptr && obs_li && obs_uv);
  if (q->job_active) (void)sfmx_ba_end(c, q);
  const int R = obs_ptr[P];
  SFMX_REQUIRE(c, R >= 0 && obs_ptr[0] == 0);
  q->W = W; q->P = P; q->R = R;
  q->MS = W < BA_MAX_OBS ? W : BA_MAX_OBS;
  const int D = 6 * W;
  const size_t CS = (size_t)36 * W * W + 36 * W + 12 * W;
  // the four input arrays share one device slab [X | obs_uv | obs_ptr | obs_li] (16-byte aligned parts) and travel
  // in ONE transfer from a pinned staging slab; no host wait: the first step is ordered behind it on the stream
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t o_x = 0, o_uv = up16((size_t)P * 24), o_ptr = o_uv + up16((size_t)R * 16), o_li = o_ptr + up16((size_t)(P + 1) * 4);
  const size_t in_bytes = o_li + up16((size_t)R * 4) + 16;
  q->chunk = ba_merged(q) ? 0 : ba_chunk_points();
  const size_t row_slots = q->chunk ? (size_t)2 * q->chunk : (size_t)P;  // all rows, or the two-slot ring of chunks
  const size_t need[11] = {in_bytes, 16, 16, 16, (size_t)W * 96,
                           (size_t)P * q->MS * BA_SLOT * 8, (size_t)P * W, (size_t)D * D * 8 + (size_t)D * 8, 16, (size_t)D * 8 + 64,
                           row_slots * CS * 8};
  const void* ticket_before = q->bufs[8].p;
  for (int i = 0; i < 11; i++) SFMX_HIP(c, q->bufs[i].ensure(need[i]));
  const bool new_ticket = q->bufs[8].p != ticket_before;  // zeroed once: every launch leaves the counter at zero
  q->ticket = q->bufs[8].as<unsigned>();
  q->contrib = q->bufs[10].as<double>();
  char* in = q->bufs[0].as<char>();
  q->X = reinterpret_cast<double*>(in + o_x); q->obs_uv = reinterpret_cast<double*>(in + o_uv);
  q->obs_ptr = reinterpret_cast<int32_t*>(in + o_ptr); q->obs_li = reinterpret_cast<int32_t*>(in + o_li);
  q->poses = q->bufs[4].as<double>(); q->rec = q->bufs[5].as<double>();
  q->slot_of = q->bufs[6].as<int8_t>(); q->S = q->bufs[7].as<double>(); q->b = q->S + (size_t)D * D;  // S | b contiguous: one all-reduce
  q->work = q->bufs[9].as<double>();
  // does any pose observe a point twice?  (the LDS points kernel leaves that read-modify-write case to the general one)
  q->lds_points = false;
  if (ba_merged(q)) {
    bool dup = false;
    for (int p = 0; p < P && !dup; p++) {
      const int o0 = obs_ptr[p], cnt = obs_ptr[p + 1] - o0;
      if (cnt < 2 || cnt > BA_MAX_OBS) continue;
      unsigned long long seen = 0;  // W <= 64
      for (int k = 0; k < cnt; k++) {
        const int li = obs_li[o0 + k];
        if (li < 0 || li >= W) continue;
        if (seen >> li & 1ull) { dup = true; break; }
        seen |= 1ull << li;
      }
    }
    q->lds_points = !dup;
  }
  if (c->ba_upload_in_flight) {  // two resets in a row: the staging slab is still being read
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
    c->ba_upload_in_flight = false;
  }
  SFMX_HIP(c, c->h[4].ensure(in_bytes));
  char* st = c->h[4].as<char>();
  memcpy(st + o_x, X, (size_t)P * 24);
  memcpy(st + o_ptr, obs_ptr, (size_t)(P + 1) * 4);
  if (R > 0) {
    memcpy(st + o_uv, obs_uv, (size_t)R * 16);
    memcpy(st + o_li, obs_li, (size_t)R * 4);
  }
  SFMX_HIP(c, hipMemcpyAsync(in, st, in_bytes, hipMemcpyHostToDevice, c->stream));
  if (new_ticket) SFMX_HIP(c, hipMemsetAsync(q->ticket, 0, 16, c->stream));
  c->ba_upload_in_flight = true;  // cleared by the first build / step, which wait for the stream
  return SFMX_OK;
}

int sfmx_ba_create(sfmx_ctx* c, int W, int P, const double* X, const int32_t* obs_ptr, const int32_t* obs_li, const double* obs_uv,
                   sfmx_ba_problem** out) {
  SFMX_REQUIRE(c, c && out);
  sfmx_ba_problem* q = new sfmx_ba_problem;
  const int rc = sfmx_ba_reset(c, q, W, P, X, obs_ptr, obs_li, obs_uv);
  if (rc != SFMX_OK) {
    sfmx_ba_destroy(c, q);
    return rc;
  }
  *out = q;
  return SFMX_OK;
}

void sfmx_ba_destroy(sfmx_ctx* c, sfmx_ba_problem* q) {
  if (!q) return;
  if (q->job_active && c) (void)sfmx_ba_end(c, q);
  if (c) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->copy_stream); }
  for (auto& b : q->bufs) b.release();
  q->job_ctl.release();
  q->job_cmd.release();
  if (q->ev_sync) (void)hipEventDestroy(q->ev_sync);
  for (int k = 0; k < 2; k++) {
    if (q->ev_exp[k]) (void)hipEventDestroy(q->ev_exp[k]);
    if (q->ev_red[k]) (void)hipEventDestroy(q->ev_red[k]);
  }
  delete q;
}

int sfmx_ba_build(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx, double cy, double huber,
                  double lambda, int damp, double* S_out, double* b_out) {
  SFMX_REQUIRE(c, c && q && poses_wc && S_out && b_out);
  KernelTimer t(c);
  int rc = ba_launch_build(c, q, poses_wc, fx, fy, cx, cy, huber, lambda, damp, t);
  if (rc) return rc;
  const int D = 6 * q->W;
  SFMX_HIP(c, hipMemcpyAsync(S_out, q->S, (size_t)D * D * 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(b_out, q->b, (size_t)D * 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  return SFMX_OK;
}

int sfmx_ba_build_partial(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx, double cy,
                          double huber, void** S_dev, void** b_dev) {
  SFMX_REQUIRE(c, c && q && poses_wc && S_dev && b_dev);
  KernelTimer t(c);
  int rc = ba_launch_build(c, q, poses_wc, fx, fy, cx, cy, huber, 0.0, 0, t);
  if (rc) return rc;
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  *S_dev = q->S;
  *b_dev = q->b;
  return SFMX_OK;
}

// csrc/hip/solve_host.cpp (g++): dense.hpp:54-93 on the host core that polls for the result
extern "C" int sfmx_host_solve_window(const double* S, const double* b, int n, double* x, double* work);

// A BA job = up to `iters` sfmx_ba_step calls on one problem with the same intrinsics.  sfmx_ba_begin launches the resident kernel
// (window-sized problems with a 36-unknown system; anything else keeps the two launches per step), sfmx_ba_step then only hands
// the poses over and polls for S | b, sfmx_ba_end releases the kernel if the job ended early (singular system).  Optional: a
// step outside a job works as before.  SFMX_BA_RESIDENT=0 switches the mode off (A/B and tests; identical results).
int sfmx_ba_begin(sfmx_ctx* c, sfmx_ba_problem* q, int iters, double fx, double fy, double cx, double cy, double huber, double lambda) {
  SFMX_REQUIRE(c, c && q && iters >= 0);
  if (q->job_active) (void)sfmx_ba_end(c, q);
  // Opt-in (SFMX_BA_RESIDENT=1): built, bit-exact and tested, but measured SLOWER inside the pipeline than two launches per
  // iteration (lane B 27.9 ms against 18.6 ms per 47-frame pass, profiles/r03_ab_ba_resident.txt) -- see DESIGN.md 4.4.
  const char* e = getenv("SFMX_BA_RESIDENT");
  const char* se = getenv("SFMX_BA_SOLVE");
  const bool off = !(e && e[0] == '1') || (se && std::string(se) == "device") || getenv("SFMX_BA_NO_POLL") || getenv("SFMX_BA_NO_FUSE");
  if (off || c->timing || iters < 1 || q->W != 6 || q->chunk != 0 || q->P > BA_MERGED_EXPAND_MAX_P) return SFMX_OK;
  const int D = 6 * q->W, NE = D * D + D;
  SFMX_HIP(c, q->job_ctl.ensure(sizeof(BaResidentCtl)));
  SFMX_HIP(c, q->job_cmd.ensure(16 * 8));
  q->job_slot = (q->job_slot + 1) & 15;
  SFMX_HIP(c, c->h[0].ensure((size_t)q->W * 96));
  SFMX_HIP(c, c->h[1].ensure((size_t)NE * 8 + 32));
  __atomic_store_n(q->job_cmd.as<unsigned long long>() + q->job_slot, 0ull, __ATOMIC_RELEASE);
  *reinterpret_cast<volatile unsigned long long*>(c->h[1].as<double>() + NE) = 0;
  SFMX_HIP(c, hipMemsetAsync(q->job_ctl.p, 0, sizeof(BaResidentCtl), c->stream));
  q->job_first_seq = c->ba_seq + 1;
  q->job_iters = iters;
  q->job_done = 0;
  const double k[6] = {fx, fy, cx, cy, huber, lambda};
  memcpy(q->job_k, k, sizeof k);
  k_ba_window_resident<<<ba_element_blocks(q), 256, 0, c->stream>>>(q->W, q->P, q->MS, c->h[0].as<double>(), q->X, q->obs_ptr, q->obs_li, q->obs_uv, fx, fy,
                                                                   cx, cy, huber, lambda, q->rec, q->slot_of, q->contrib, q->S, q->b,
                                                                   q->job_ctl.as<BaResidentCtl>(), q->job_cmd.as<unsigned long long>() + q->job_slot,
                                                                   c->h[1].as<double>(), q->job_first_seq, iters, ba_wave_prio(), ba_element_blocks(q));
  SFMX_HIP(c, hipGetLastError());
  q->job_active = true;
  c->ba_upload_in_flight = false;  // the kernel is ordered behind the upload on the stream; nothing else touches the staging slab
  return SFMX_OK;
}
int sfmx_ba_end(sfmx_ctx* c, sfmx_ba_problem* q) {
  SFMX_REQUIRE(c, c && q);
  if (!q->job_active) return SFMX_OK;
  if (q->job_done < q->job_iters)  // ended early: the kernel is waiting for a command
    __atomic_store_n(q->job_cmd.as<unsigned long long>() + q->job_slot, BAW_EXIT, __ATOMIC_RELEASE);
  q->job_active = false;
  return SFMX_OK;
}
// one step of an active job: poses -> command word -> S | b -> host solve
static int ba_step_resident(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, double* dx_out) {
  const int D = 6 * q->W, NE = D * D + D;
  const unsigned long long seq = ++c->ba_seq;  // == job_first_seq + job_done
  double* hout = c->h[1].as<double>();
  memcpy(c->h[0].p, poses_wc, (size_t)q->W * 96);
  __atomic_store_n(q->job_cmd.as<unsigned long long>() + q->job_slot, seq, __ATOMIC_RELEASE);
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(hout + NE);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(500)) {
      __atomic_store_n(q->job_cmd.as<unsigned long long>() + q->job_slot, BAW_EXIT, __ATOMIC_RELEASE);
      q->job_active = false;
      SFMX_HIP(c, hipStreamSynchronize(c->stream));
      return sfmx_fail(c, SFMX_ERR_HIP, "BA step result never arrived (resident kernel)", hipSuccess);
    }
  }
  q->job_done++;
  if (q->job_done >= q->job_iters) q->job_active = false;  // the kernel leaves by itself after its last iteration
  double work[36 * 37];
  const int status = sfmx_host_solve_window(hout, hout + (size_t)D * D, D, dx_out, work);
  return status ? SFMX_ERR_SINGULAR : SFMX_OK;
}

int sfmx_ba_step(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx, double cy, double huber,
                 double lambda, double* dx_out) {
  SFMX_REQUIRE(c, c && q && poses_wc && dx_out);
  if (q->job_active) {
    const double k[6] = {fx, fy, cx, cy, huber, lambda};
    if (memcmp(k, q->job_k, sizeof k) == 0 && !c->timing) return ba_step_resident(c, q, poses_wc, dx_out);
    (void)sfmx_ba_end(c, q);  // other parameters than the job was started with: release it, take the plain path
  }
  const int D = 6 * q->W, NE = D * D + D;
  static const bool no_poll = getenv("SFMX_BA_NO_POLL") != nullptr;
  static const bool no_fuse = getenv("SFMX_BA_NO_FUSE") != nullptr;  // A/B and tests: reduce, solve and publish as separate launches
  // Where the window's system is solved: on the polling host core by default (3-4 us; the elimination is a chain of dependent
  // steps that one wavefront needs ~30 us for), SFMX_BA_SOLVE=device keeps it in the last workgroup of the reduction.
  const char* solve_env = getenv("SFMX_BA_SOLVE");  // read per call: the tests switch it inside one process
  const bool host_solve_on = !(solve_env && std::string(solve_env) == "device");
  const bool poll = !c->timing && !no_poll;  // the event timers need the stream synchronisation
  const bool fuse = ba_can_fuse_solve(q) && !no_fuse;
  const bool host_solve = host_solve_on && poll && fuse;
  // SFMX_BA_HOST_TIMES=1 (diagnostic): where a step's wall time goes on the host -- launches, wait for the published system, solve,
  // and the time between two steps (the caller's pose update) -- averaged over 1 000 steps, on stderr
  static const bool host_times = getenv("SFMX_BA_HOST_TIMES") != nullptr;
  static double ht_launch = 0, ht_wait = 0, ht_solve = 0, ht_between = 0;  // (one BA lane at a time; a diagnostic)
  static unsigned ht_n = 0;
  static std::chrono::steady_clock::time_point ht_last_exit;
  const auto ht0 = std::chrono::steady_clock::now();
  auto ht_us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  if (host_times && ht_n > 0 && ht_us(ht_last_exit, ht0) < 200.0) ht_between += ht_us(ht_last_exit, ht0);
  KernelTimer t(c);
  SFMX_HIP(c, c->h[1].ensure(host_solve ? (size_t)NE * 8 + 32 : (size_t)D * 8 + 32));
  double* hout = c->h[1].as<double>();
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(host_solve ? hout + NE : hout + D + 1);
  const unsigned long long seq = ++c->ba_seq;
  if (poll) *flag = 0;  // (a freshly grown buffer holds arbitrary bytes)
  // SFMX_BA_PUBLISH=last: S | b copied to the pinned block by the last workgroup of the reduction (A/B and tests) instead of by every
  // workgroup for its own elements
  const char* pub_env = getenv("SFMX_BA_PUBLISH");
  const int publish_mode = (pub_env && pub_env[0] == 'l') ? 1 : 2;
  int rc = ba_launch_build(c, q, poses_wc, fx, fy, cx, cy, huber, lambda, 1, t, poll, fuse, poll ? hout : nullptr, seq, host_solve ? publish_mode : 0);
  if (rc) return rc;
  if (!fuse) {
    int* dstatus = reinterpret_cast<int*>(q->work + D);
    rc = launch_solve(c, q->S, q->b, D, q->work, dstatus);
    if (rc) return rc;
  }
  int status = 0;
  if (poll) {
    if (!fuse) {
      k_ba_publish<<<1, 64, 0, c->stream>>>(q->work, D, hout, seq);
      SFMX_HIP(c, hipGetLastError());
    }
    const auto t0 = std::chrono::steady_clock::now();
    if (host_times) ht_launch += ht_us(ht0, t0);
    unsigned spins = 0;
    while (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq) {
      __builtin_ia32_pause();
      if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
        SFMX_HIP(c, hipStreamSynchronize(c->stream));  // an error (or a very slow device): let the runtime report it
        if (__atomic_load_n(const_cast<unsigned long long*>(flag), __ATOMIC_ACQUIRE) != seq)
          return sfmx_fail(c, SFMX_ERR_HIP, "BA step result never arrived", hipSuccess);
      }
    }
    c->ba_upload_in_flight = false;  // everything queued before the solve has completed
    if (host_solve) {
      double work[60 * 61];
      static_assert(BA_MAX_W >= 10, "window sizes with a fused reduction: 6 and 10 poses");
      const auto ts = std::chrono::steady_clock::now();
      status = sfmx_host_solve_window(hout, hout + (size_t)D * D, D, dx_out, work);
      if (host_times) {
        const auto te = std::chrono::steady_clock::now();
        ht_wait += ht_us(t0, ts);
        ht_solve += ht_us(ts, te);
        ht_last_exit = te;
        if (++ht_n == 1000) {
          fprintf(stderr, "[sfmx] ba step host times (us, mean of 1000): launches %.1f | wait for S|b %.1f | host solve %.1f | between steps %.1f\n", ht_launch / 1000,
                  ht_wait / 1000, ht_solve / 1000, ht_between / 1000);
          ht_launch = ht_wait = ht_solve = ht_between = 0;
          ht_n = 0;
        }
      }
      return status ? SFMX_ERR_SINGULAR : SFMX_OK;
    }
    memcpy(dx_out, hout, (size_t)D * 8);
    memcpy(&status, hout + D, 4);
    return status ? SFMX_ERR_SINGULAR : SFMX_OK;
  }
  SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, q->work, (size_t)D * 8 + 4, hipMemcpyDeviceToHost, c->stream));  // dx | status
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  memcpy(dx_out, c->h[1].p, (size_t)D * 8);
  memcpy(&status, c->h[1].as<char>() + (size_t)D * 8, 4);
  return status ? SFMX_ERR_SINGULAR : SFMX_OK;
}

int sfmx_ba_step_sharded(sfmx_ctx* c, sfmx_comm* comm, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx, double cy,
                         double huber, double lambda, double* dx_out) {
  SFMX_REQUIRE(c, c && q && poses_wc && dx_out);
  const int D = 6 * q->W;
  KernelTimer t(c);
  const int vworld = (!comm || comm->world <= 1) ? ba_virtual_world() : 0;     // test mode, see ba_build_virtual_world
  int rc = (vworld > 1 && q->P >= vworld) ? ba_build_virtual_world(c, q, vworld, poses_wc, fx, fy, cx, cy, huber, t)
                                          : ba_launch_build(c, q, poses_wc, fx, fy, cx, cy, huber, 0.0, 0, t);  // raw sums of this rank's points
  if (rc) return rc;
  rc = sfmx_comm_allreduce_dev(c, comm, q->S, (size_t)D * D + D, 0, 0);        // S | b, in HBM, on the BA stream
  if (rc) return rc;
  k_ba_damp_gauge<<<(D + 63) / 64, 64, 0, c->stream>>>(q->S, q->b, D, lambda);
  int* dstatus = reinterpret_cast<int*>(q->work + D);
  rc = launch_solve(c, q->S, q->b, D, q->work, dstatus);
  if (rc) return rc;
  int status = 0;
  SFMX_HIP(c, c->h[1].ensure((size_t)D * 8 + 8));
  SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, q->work, (size_t)D * 8 + 4, hipMemcpyDeviceToHost, c->stream));  // dx | status
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  memcpy(dx_out, c->h[1].p, (size_t)D * 8);
  memcpy(&status, c->h[1].as<char>() + (size_t)D * 8, 4);
  return status ? SFMX_ERR_SINGULAR : SFMX_OK;
}

// Element-sharded BA iteration -- the sharded mode that keeps the reference's arithmetic.  Every rank holds the WHOLE window and
// computes the per-point records of all points (replicated: 10 % of an iteration at C4 size); rank r then forms and reduces only
// ITS contiguous slice of the element blocks of S | b, each element's chain over all points in reference order, and contributes
// +0.0 everywhere else.  The all-reduce(sum) adds zeros to every element (x + 0.0 == x; the running sums are never -0.0), so the
// system -- and dx -- is bit-identical to sfmx_ba_step on one GPU at any world size and for any reduction order RCCL picks.
// (Point sharding, sfmx_ba_step_sharded, regroups the addends of every element; on this BA that rounding difference is
// amplified to a different trajectory within tens of keyframes -- tools/virtual_world_probe.py)
int sfmx_ba_step_sharded_elements(sfmx_ctx* c, sfmx_comm* comm, sfmx_ba_problem* q, const double* poses_wc, double fx, double fy, double cx,
                                  double cy, double huber, double lambda, double* dx_out) {
  SFMX_REQUIRE(c, c && q && poses_wc && dx_out);
  const int D = 6 * q->W, NE = D * D + D;
  const bool real = comm && comm->world > 1;
  const int vworld = real ? 0 : ba_virtual_world();  // test mode: the N slices formed one after the other on this GPU
  const int world = real ? comm->world : (vworld > 1 ? vworld : 1);
  const int nblk = ba_element_blocks(q);
  KernelTimer t(c);
  int rc = SFMX_OK;
  const double* d_poses = ba_stage_poses(c, q, poses_wc, false, &rc);
  if (rc) return rc;
  double* part = nullptr;
  if (vworld > 1) {
    SFMX_HIP(c, q->bufs[11].ensure((size_t)vworld * NE * 8));
    part = q->bufs[11].as<double>();
    SFMX_HIP(c, hipMemsetAsync(part, 0, (size_t)vworld * NE * 8, c->stream));
  }
  t.start();
  for (int r = 0; r < (vworld > 1 ? vworld : 1); ++r) {
    const int rank = real ? comm->rank : r;
    int b_lo = 0, b_hi = nblk;
    sfmx_shard_range(nblk, rank, world, &b_lo, &b_hi);
    const int e_lo = b_lo * BAR_COLS, e_hi = b_hi * BAR_COLS < NE ? b_hi * BAR_COLS : NE;
    double* S_out = part ? part + (size_t)r * NE : q->S;
    if (!part && world > 1) SFMX_HIP(c, hipMemsetAsync(q->S, 0, (size_t)NE * 8, c->stream));  // S | b contiguous: +0.0 outside the slice
    if (r == 0) {  // the records do not depend on the slice
      rc = ba_launch_points(c, q, d_poses, fx, fy, cx, cy, huber, ba_wave_prio());
      if (rc) return rc;
    }
    rc = ba_launch_reduce(c, q, 0, q->P, 0.0, 0, S_out, S_out + (size_t)D * D, nullptr, false, nullptr, 0, ba_wave_prio(), b_lo, b_hi - b_lo, 0, e_lo, e_hi);
    if (rc) return rc;
  }
  if (part) {
    const char* oe = getenv("SFMX_VIRTUAL_WORLD_ORDER");
    const std::string order_s = oe ? oe : "rank";
    const int order = order_s == "reverse" ? 1 : order_s == "ring" ? 2 : order_s == "tree" ? 3 : 0;
    k_ba_combine_partials<<<(NE + 255) / 256, 256, 0, c->stream>>>(part, vworld, NE, order, q->S);
  }
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  rc = sfmx_comm_allreduce_dev(c, comm, q->S, (size_t)NE, 0, 0);  // S | b, in HBM, on the BA stream
  if (rc) return rc;
  k_ba_damp_gauge<<<(D + 63) / 64, 64, 0, c->stream>>>(q->S, q->b, D, lambda);
  int* dstatus = reinterpret_cast<int*>(q->work + D);
  rc = launch_solve(c, q->S, q->b, D, q->work, dstatus);
  if (rc) return rc;
  int status = 0;
  SFMX_HIP(c, c->h[1].ensure((size_t)D * 8 + 8));
  SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, q->work, (size_t)D * 8 + 4, hipMemcpyDeviceToHost, c->stream));  // dx | status
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  memcpy(dx_out, c->h[1].p, (size_t)D * 8);
  memcpy(&status, c->h[1].as<char>() + (size_t)D * 8, 4);
  return status ? SFMX_ERR_SINGULAR : SFMX_OK;
}

int sfmx_solve_dense(sfmx_ctx* c, const double* A, const double* b, int n, double* x) {
  SFMX_REQUIRE(c, c && A && b && x && n >= 1);
  const size_t nb = (size_t)n * n * 8;
  c->resident_points = 0;
  SFMX_HIP(c, c->d[0].ensure(nb));
  SFMX_HIP(c, c->d[1].ensure((size_t)n * 8));
  SFMX_HIP(c, c->d[2].ensure((size_t)n * 8 + 64));
  SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, A, nb, hipMemcpyHostToDevice, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(c->d[1].p, b, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  int* dstatus = reinterpret_cast<int*>(c->d[2].as<double>() + n);
  KernelTimer t(c);
  t.start();
  int rc = launch_solve(c, c->d[0].as<double>(), c->d[1].as<double>(), n, c->d[2].as<double>(), dstatus);
  t.stop();
  if (rc) return rc;
  int status = 0;
  SFMX_HIP(c, hipMemcpyAsync(x, c->d[2].p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(&status, dstatus, 4, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  c->ba_upload_in_flight = false;
  t.collect();
  return status ? SFMX_ERR_SINGULAR : SFMX_OK;
}

}  // extern "C"
