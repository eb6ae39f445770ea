// ransac.hip — 8-point hypothesis generation and Sampson-error scoring for gfx950.
//
// Replaces the hypothesis loop of find_E_ransac (reference cpp/src/templering_sfm.cpp T:664-677):
// eight_point_E (T:609-627: 8x9 design matrix, AtA, 9x9 Jacobi, rank-2 projection through svd3)
// and the per-point sampson_err test (T:629-638, T:669-672).
//
// Parity contract: inlier COUNTS and MASKS are integer results and must equal
// the reference's.  The Sampson kernels evaluate T:629-638 operation for operation (no FMA), so
// for a given E the mask is bit-exact.  Hypothesis generation contains the reference's only
// libm-dependent step on this path (phi = 0.5*atan2(2apq, aqq-app); cos/sin, linalg.hpp:156-157),
// which no device library reproduces bit-for-bit; the device therefore forms the same Jacobi
// rotation algebraically (half-angle identities, <= 2 ulp from the libm values) and its E's are
// used ONLY to rank hypotheses.  The host re-derives the winner's E with the platform libm and the
// final mask is recomputed from that E by k_sampson_mask, so everything that leaves the library is
// bit-identical to the reference as long as the ranking agrees (tests/ verifies on all fixtures).
//
// k_hypotheses: 16 lanes per hypothesis, 4 hypotheses per wavefront, matrices in LDS; the pivot
//   search and the row / column / eigenvector rotations are spread over the lanes and reduced with
//   16-lane shuffles (first-maximum tie-break = the reference's row-major strict '>' scan).
// k_score: 256 threads x 8 hypotheses per block; each point (32 B, coalesced from L2) is loaded once
//   per block and tested against the block's 8 essential matrices, which sit in SGPRs (uniform loads).
#include "sfmx_internal.h"

#include <cstdlib>
#include <cstring>
#include <vector>

#define HG 16  // lanes per hypothesis
#define HPW 4  // hypotheses per wave
// A Jacobi pivot is "nearly tied" when a second off-diagonal entry lies within this relative band of the
// maximum: rounding differences between this kernel and the reference could then pick different pivots, i.e.
// follow a different rotation sequence.  Such hypotheses are reported with conditioning 0 (=> exact host E).
#define PIVOT_TIE_BAND 1e-9

struct Rot { double c, s; };
// Jacobi rotation of linalg.hpp:153-157 without libm: phi = atan2(y, x)/2 in (-pi/2, pi/2].
__device__ __forceinline__ Rot half_angle(double y, double x) {
  Rot r;
  const double h = sqrt(y * y + x * x);
  if (!(h > 0.0)) {
    r.c = 1.0;
    r.s = 0.0;
    return r;
  }
  const double c2 = x / h, s2 = y / h;
  if (x >= 0.0) {
    r.c = sqrt(0.5 * (1.0 + c2));
    r.s = s2 / (2.0 * r.c);
  } else {
    const double sa = sqrt(0.5 * (1.0 - c2));
    r.s = (y < 0.0) ? -sa : sa;
    r.c = fabs(s2) / (2.0 * sa);
  }
  return r;
}

// Same rotation without IEEE divisions / square roots: v_rsq_f64 seed + one Newton step (relative
// error ~1e-15).  Only used for the 9x9 hypothesis Jacobi, whose E's rank hypotheses and never leave
// the library (the winner is re-derived on the host); Jacobi is self-correcting, so the final
// eigenvectors are as accurate as with the exact rotation.
__device__ __forceinline__ double rsqrt_nr(double a) {
  double r = __builtin_amdgcn_rsq(a);
  r = r * (1.5 - 0.5 * a * r * r);
  r = r * (1.5 - 0.5 * a * r * r);
  return r;
}
__device__ __forceinline__ Rot half_angle_fast(double y, double x) {
  // branch-free: the four hypotheses of a wave take different sign cases, and a divergent branch runs both rsqrt chains
  Rot r;
  const double d = y * y + x * x;
  const double rh = rsqrt_nr(d);
  const double c2 = x * rh, s2 = y * rh;
  const bool pos = x >= 0.0;
  const double u = 0.5 * (1.0 + (pos ? c2 : -c2)), ru = rsqrt_nr(u);
  const double m = u * ru;                                    // sqrt(u)
  const double hs = 0.5 * (pos ? s2 : fabs(s2)) * ru;
  r.c = pos ? m : hs;
  r.s = pos ? hs : ((y < 0.0) ? -m : m);
  if (!(d > 0.0) || !(d < 1e300)) return half_angle(y, x);  // rare: zero / huge / NaN pivots
  return r;
}

// serial 3x3 Jacobi (linalg.hpp:133-201, N=3) on LDS-resident A3/V3, executed by one lane
// returns true if some pivot choice was a near tie (see PIVOT_TIE_BAND)
__device__ bool jacobi3_lds(double* A, double* V, int sweeps) {
  bool near_tie = false;
  for (int i = 0; i < 9; i++) V[i] = 0.0;
  V[0] = V[4] = V[8] = 1.0;
  for (int it = 0; it < sweeps; ++it) {
    int p = 0, q = 1;
    double big = 0.0;
    const double a01 = fabs(A[1]), a02 = fabs(A[2]), a12 = fabs(A[5]);
    if (a01 > big) { big = a01; p = 0; q = 1; }
    if (a02 > big) { big = a02; p = 0; q = 2; }
    if (a12 > big) { big = a12; p = 1; q = 2; }
    if (big < 1e-12) break;
    const double band = big * (1.0 - PIVOT_TIE_BAND);
    near_tie |= ((a01 >= band) + (a02 >= band) + (a12 >= band)) > 1;
    const Rot r = half_angle(2.0 * A[p * 3 + q], A[q * 3 + q] - A[p * 3 + p]);
    const double c = r.c, s = r.s;
    for (int k = 0; k < 3; k++) {
      const double ap = A[p * 3 + k], aq = A[q * 3 + k];
      A[p * 3 + k] = c * ap - s * aq;
      A[q * 3 + k] = s * ap + c * aq;
    }
    for (int k = 0; k < 3; k++) {
      const double ap = A[k * 3 + p], aq = A[k * 3 + q];
      A[k * 3 + p] = c * ap - s * aq;
      A[k * 3 + q] = s * ap + c * aq;
    }
    A[p * 3 + q] = 0.0;
    A[q * 3 + p] = 0.0;
    for (int k = 0; k < 3; k++) {
      const double vp = V[k * 3 + p], vq = V[k * 3 + q];
      V[k * 3 + p] = c * vp - s * vq;
      V[k * 3 + q] = s * vp + c * vq;
    }
  }
  return near_tie;
}

__device__ __forceinline__ void unit3(double& x, double& y, double& z) {
  const double n = sqrt(x * x + y * y + z * z);
  if (!isfinite(n) || n < 1e-12) { x = y = z = 0.0; return; }
  x = x / n; y = y / n; z = z / n;
}

// T:537-607: E -> U diag(s0,s1,0) V^T, executed by one lane; E9 in/out in LDS, scratch: A3,V3 (LDS)
// returns the conditioning of the projection: (s1^2 - s2^2) / s0^2 (0 when a 3x3 pivot was nearly tied)
__device__ double rank2_project(double* E, double* A3, double* V3) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += E[3 * k + r] * E[3 * k + c];
      A3[3 * r + c] = acc;
    }
  const bool tie3 = jacobi3_lds(A3, V3, 80);
  double w[3] = {A3[0], A3[4], A3[8]};
  // eigenvalues ascending (insertion sort as libstdc++ does for N<=16), then singular values descending
  int pe[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++) {
    const int val = pe[i];
    if (w[val] < w[pe[0]]) { for (int k = i; k > 0; k--) pe[k] = pe[k - 1]; pe[0] = val; }
    else { int k = i; while (w[val] < w[pe[k - 1]]) { pe[k] = pe[k - 1]; k--; } pe[k] = val; }
  }
  double sv[3];
  for (int i = 0; i < 3; i++) { const double wi = w[pe[i]]; sv[i] = sqrt((0.0 < wi) ? wi : 0.0); }
  int od[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++) {
    const int val = od[i];
    if (sv[val] > sv[od[0]]) { for (int k = i; k > 0; k--) od[k] = od[k - 1]; od[0] = val; }
    else { int k = i; while (sv[val] > sv[od[k - 1]]) { od[k] = od[k - 1]; k--; } od[k] = val; }
  }
  double Vd[9], sd[3];
  for (int c = 0; c < 3; c++) {
    sd[c] = sv[od[c]];
    const int src = pe[od[c]];
    for (int r = 0; r < 3; r++) Vd[3 * r + c] = V3[3 * r + src];
  }
  double U[9];
  for (int c = 0; c < 3; c++) {
    const double vx = Vd[c], vy = Vd[3 + c], vz = Vd[6 + c];
    double ux = E[0] * vx + E[1] * vy + E[2] * vz;
    double uy = E[3] * vx + E[4] * vy + E[5] * vz;
    double uz = E[6] * vx + E[7] * vy + E[8] * vz;
    if (sd[c] > 1e-12) { ux = ux / sd[c]; uy = uy / sd[c]; uz = uz / sd[c]; }
    else unit3(ux, uy, uz);
    U[c] = ux; U[3 + c] = uy; U[6 + c] = uz;
  }
  double u0x = U[0], u0y = U[3], u0z = U[6], u1x = U[1], u1y = U[4], u1z = U[7];
  unit3(u0x, u0y, u0z);
  const double d01 = u0x * u1x + u0y * u1y + u0z * u1z;
  u1x = u1x - d01 * u0x; u1y = u1y - d01 * u0y; u1z = u1z - d01 * u0z;
  unit3(u1x, u1y, u1z);
  double u2x = u0y * u1z - u0z * u1y, u2y = u0z * u1x - u0x * u1z, u2z = u0x * u1y - u0y * u1x;
  unit3(u2x, u2y, u2z);
  U[0] = u0x; U[3] = u0y; U[6] = u0z; U[1] = u1x; U[4] = u1y; U[7] = u1z; U[2] = u2x; U[5] = u2y; U[8] = u2z;
  // US = U * diag(s0,s1,0) with the reference's full 3-term sums, then (US) * V^T
  const double S[9] = {sd[0], 0, 0, 0, sd[1], 0, 0, 0, 0.0};
  double US[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += U[3 * r + k] * S[3 * k + c];
      US[3 * r + c] = acc;
    }
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double acc = 0.0;
      for (int k = 0; k < 3; k++) acc += US[3 * r + k] * Vd[3 * c + k];  // Vt(k,c) = V(c,k)
      E[3 * r + c] = acc;
    }
  const double cond = (sd[1] * sd[1] - sd[2] * sd[2]) / (sd[0] * sd[0]);
  return (tie3 || !(cond > 0.0)) ? 0.0 : cond;
}

// one step of the 16-lane (value, code) arg-max all-reduce used by the Jacobi pivot search
template <int CTRL>
__device__ __forceinline__ void pivot_combine(double& v, int& code) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  const int oc = __builtin_amdgcn_update_dpp(0, code, CTRL, 0xF, 0xF, false);
  const double ov = __hiloint2double(hi, lo);
  const bool take = (ov > v) | ((ov == v) & (oc < code));
  v = take ? ov : v;
  code = take ? oc : code;
}

template <int CTRL>
__device__ __forceinline__ void pivot_combine_key(unsigned long long& k) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)k, CTRL, 0xF, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(k >> 32), CTRL, 0xF, 0xF, false);
  const unsigned long long o = ((unsigned long long)hi << 32) | lo;
  k = o > k ? o : k;
}

// 16-lane all-reduce maximum of an unsigned word: four v_max_u32 with a DPP operand (0 is the identity of the unsigned maximum,
// so the cross-lane move folds into the instruction)
__device__ __forceinline__ unsigned row_umax(unsigned x) {
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false));  // row_half_mirror
  x = max(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false));  // row_mirror
  return x;
}

struct HypLds {
  double D[72];   // 8x9 design matrix
  double A[81];   // AtA, rotated in place
  double V[81];   // eigenvectors (columns)
  double E[9];
  double A3[9], V3[9];
};

template <bool LEAN, bool STAMP = false>
__global__ __launch_bounds__(64) void k_hypotheses(const double* __restrict__ xi, const double* __restrict__ xj, int n,
                                                   const int32_t* __restrict__ idx8, int H, int sweeps, double* __restrict__ E_out,
                                                   double* __restrict__ cond_out, unsigned long long* __restrict__ ticks) {
  __shared__ HypLds lds[HPW];
  const unsigned long long t_kernel = ticks ? __builtin_amdgcn_s_memtime() : 0ull;
  const int lane = threadIdx.x, g = lane / HG, t = lane % HG;
  const int hyp = blockIdx.x * HPW + g;
  const bool live = hyp < H;
  HypLds& L = lds[g];
  // ---- design matrix rows (T:612-618)
  if (live && t < 8) {
    int i = idx8[(size_t)hyp * 8 + t];
    i = min(max(i, 0), n - 1);
    const double x = xi[2 * i], y = xi[2 * i + 1], xp = xj[2 * i], yp = xj[2 * i + 1];
    double* row = L.D + 9 * t;
    row[0] = xp * x; row[1] = xp * y; row[2] = xp;
    row[3] = yp * x; row[4] = yp * y; row[5] = yp;
    row[6] = x; row[7] = y; row[8] = 1.0;
  }
  __syncthreads();
  // ---- AtA upper triangle, mirrored (T:503-517): 45 entries over 16 lanes
  if (live) {
    for (int e = t; e < 45; e += HG) {
      int i = 0, rem = e;
      while (rem >= 9 - i) { rem -= 9 - i; i++; }
      const int j = i + rem;
      double acc = 0.0;
      for (int r = 0; r < 8; r++) acc += L.D[r * 9 + i] * L.D[r * 9 + j];
      L.A[i * 9 + j] = acc;
      L.A[j * 9 + i] = acc;
    }
    for (int e = t; e < 81; e += HG) L.V[e] = (e / 9 == e % 9) ? 1.0 : 0.0;
  }
  __syncthreads();
  // ---- 9x9 Jacobi (linalg.hpp:141-186): same pivot rule (first maximum of |a_ij|, i<j, row-major), same
  // rotation angle, same sweep cap.  The similarity transform A <- J^T A J is applied in its symmetric
  // one-pass form (off-diagonal pairs rotated and mirrored, 2x2 pivot block in closed form) instead of the
  // reference's row pass followed by a column pass: mathematically identical, two LDS round trips per
  // rotation instead of five.  These E's only RANK hypotheses (the winner is re-derived with libm on the
  // host), so last-bit differences from the reference's evaluation order are irrelevant here.
  // Pivot search: lane t of the group owns upper-triangle entries t, t+16, t+32 (row-major numbering) and the 16
  // lanes combine (|a|, code = 16*i + j) with a DPP all-reduce: larger |a| wins, equal |a| -> smaller code, which
  // is exactly the reference's row-major scan with strict '>' (NaN never wins; an all-zero matrix ends the loop).
  int own_off[3], own_code[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    int e = t + 16 * k, i = 0;
    if (e > 35) e = 35;
    while (e >= 8 - i) { e -= 8 - i; i++; }
    const int j = i + 1 + e;
    own_off[k] = i * 9 + j;
    own_code[k] = 16 * i + j;
  }
  const bool third = t < 4;
  unsigned long long t_begin = 0, tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
  unsigned int n_rot = 0;
  if (ticks) t_begin = __builtin_amdgcn_s_memtime();
  auto stamp = [&](int k) {  // STAMP build only: s_memtime deltas per phase of the rotation loop
    if (STAMP) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tph[k] += now - tlast;
      tlast = now;
    }
  };
  if (STAMP) tlast = t_begin;
  bool active = live;
  bool near_tie = false;  // uniform within the 16-lane group (LEAN: per lane until the loop has ended)
  const unsigned long long gmask = 0xffffull << (HG * g);  // this hypothesis' lanes in a wave-wide ballot
  for (int it = 0; it < sweeps; ++it) {
    // ---- pivot: arg-max of |a_ij| over the upper triangle, first in row-major order among equals.  The 16 lanes reduce
    // ONE 64-bit key per entry -- the magnitude's bit pattern with its low byte replaced by 255 - code -- with integer
    // compares (a (value, code) pair costs two FP64 compares per step).  Magnitudes that differ only in the byte that
    // was given up are nearly tied by any standard: such a hypothesis is flagged below and re-derived on the host.
    double v0, v1, v2;
    unsigned long long kb;
    if constexpr (LEAN) {
      // (the third read is unconditional -- its address is a valid entry for every lane -- and selected afterwards: a branch around
      // it costs an LDS wait of its own.  A NaN magnitude is not filtered: its bit pattern wins the maximum, `bv` below is NaN and
      // the hypothesis goes to the host, which is where a matrix with a NaN belongs.)
      const double a2 = L.A[own_off[2]];
      v0 = fabs(L.A[own_off[0]]);
      v1 = fabs(L.A[own_off[1]]);
      v2 = third ? fabs(a2) : 0.0;
      const unsigned long long k0 = ((unsigned long long)__double_as_longlong(v0) & ~0xffull) | (unsigned)(255 - own_code[0]);
      const unsigned long long k1 = ((unsigned long long)__double_as_longlong(v1) & ~0xffull) | (unsigned)(255 - own_code[1]);
      const unsigned long long k2 = ((unsigned long long)__double_as_longlong(v2) & ~0xffull) | (unsigned)(255 - own_code[2]);
      kb = k0 > k1 ? k0 : k1;
      kb = kb > k2 ? kb : k2;
    } else {
      v0 = fabs(L.A[own_off[0]]);
      v1 = fabs(L.A[own_off[1]]);
      v2 = third ? fabs(L.A[own_off[2]]) : 0.0;
      const unsigned long long k0 = (((v0 == v0) ? (unsigned long long)__double_as_longlong(v0) : 0ull) & ~0xffull) | (unsigned)(255 - own_code[0]);
      const unsigned long long k1 = (((v1 == v1) ? (unsigned long long)__double_as_longlong(v1) : 0ull) & ~0xffull) | (unsigned)(255 - own_code[1]);
      const unsigned long long k2 = (((v2 == v2) ? (unsigned long long)__double_as_longlong(v2) : 0ull) & ~0xffull) | (unsigned)(255 - own_code[2]);
      kb = k0 > k1 ? k0 : k1;
      kb = kb > k2 ? kb : k2;
    }
    stamp(0);  // own entries + keys
    // the 64-bit maximum in two 32-bit passes (ten instructions instead of twenty-eight for four 64-bit compare / select steps):
    // the high words first, then the low words of the lanes that hold the maximal high word
    {
      const unsigned khi = (unsigned)(kb >> 32), klo = (unsigned)kb;
      const unsigned mhi = row_umax(khi);
      const unsigned mlo = row_umax(khi == mhi ? klo : 0u);
      kb = ((unsigned long long)mhi << 32) | mlo;
    }
    const int code = 255 - (int)(kb & 0xffull);
    const double bv = __longlong_as_double((long long)(kb & ~0xffull));  // the maximum, up to its last byte
    const int p = code >> 4, q = code & 15;
    if constexpr (LEAN) {
      // maxv < 1e-12 -> break (linalg.hpp:150).  Every lane of the group holds the group's maximum up to its last byte: bv <= max <=
      // bvh.  Both on one side of 1e-12 decide the test without another reduction; straddling it (or NaN) cannot be decided from
      // the key -- that hypothesis is flagged and re-derived on the host.
      const double bvh = __longlong_as_double((long long)(kb | 0xffull));
      const bool ge = bv >= 1e-12, lt = bvh < 1e-12;
      near_tie |= active & !(ge | lt);
      active = active & ge;
      if (!__any(active)) break;
      // a second entry within PIVOT_TIE_BAND of the chosen pivot?  Flag kept per lane, combined over the group after the loop
      const double band = bv * (1.0 - PIVOT_TIE_BAND);
      near_tie |= active & (((v0 >= band) & (own_code[0] != code)) | ((v1 >= band) & (own_code[1] != code)) |
                            (third & (v2 >= band) & (own_code[2] != code)));
    } else {
    // maxv < 1e-12 -> break (linalg.hpp:150), decided on the exact magnitudes: no entry of this hypothesis reaches 1e-12
    const bool big = (v0 >= 1e-12) | (v1 >= 1e-12) | (v2 >= 1e-12);
    active = active & ((__ballot(big) & gmask) != 0);
    if (!__any(active)) break;
    {  // a second entry within PIVOT_TIE_BAND of the chosen pivot?
      const double band = bv * (1.0 - PIVOT_TIE_BAND);
      const bool mine = active & (((v0 >= band) & (own_code[0] != code)) | ((v1 >= band) & (own_code[1] != code)) |
                                  (third & (v2 >= band) & (own_code[2] != code)));
      near_tie |= (__ballot(mine) & gmask) != 0;
    }
    }
    ++n_rot;
    stamp(1);  // reduction, stop test, near-tie test
    if (active) {
      // all LDS reads of this rotation are issued together, ahead of the rotation-angle arithmetic
      const int tr = t < 9 ? t : 8;
      const double app = L.A[p * 9 + p], aqq = L.A[q * 9 + q], apq = L.A[p * 9 + q];
      const double vp = L.V[tr * 9 + p], vq = L.V[tr * 9 + q];
      const double akp = L.A[tr * 9 + p], akq = L.A[tr * 9 + q];
      if (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      stamp(2);  // LDS reads of the rotation
      const Rot r = half_angle_fast(2.0 * apq, aqq - app);
      const double c = r.c, s = r.s;
      if (STAMP) { asm volatile("" :: "v"(c), "v"(s)); }
      stamp(3);  // rotation angle
      // every lane forms the off-diagonal pair of its row and the closed-form 2x2 pivot block
      // (J = [[c, s], [-s, c]] on (p,q): a_pp' = c^2 app - 2cs apq + s^2 aqq, a_qq' = s^2 app + 2cs apq + c^2 aqq);
      // which of them a lane stores is a select, not a branch (the four hypotheses of a wave have different p, q)
      const double nvp = c * vp - s * vq, nvq = s * vp + c * vq;
      const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
      const double cc = c * c, ss = s * s, cs2 = 2.0 * c * s * apq;
      const double dpp = cc * app - cs2 + ss * aqq, dqq = ss * app + cs2 + cc * aqq;
      if (t < 9) {
        const bool isp = t == p, isq = t == q;
        const double w_tp = isp ? dpp : (isq ? 0.0 : np_);  // (t,p) and (p,t); row q: the zeroed pivot entry
        const double w_tq = isq ? dqq : (isp ? 0.0 : nq_);  // (t,q) and (q,t); row p: the zeroed pivot entry
        L.V[t * 9 + p] = nvp;
        L.V[t * 9 + q] = nvq;
        L.A[t * 9 + p] = w_tp; L.A[p * 9 + t] = w_tp;
        L.A[t * 9 + q] = w_tq; L.A[q * 9 + t] = w_tq;
      }
    }
    stamp(4);  // rotation arithmetic, stores issued
    __syncthreads();
    stamp(5);  // stores landed, barrier
  }
  if (LEAN) near_tie = (__ballot(near_tie) & gmask) != 0;  // the lanes' flags -> the hypothesis' flag
  if (ticks && lane == 0) {  // diagnostic (SFMX_RANSAC_TICKS=1|2): s_memtime ticks and rotations of this wave's loop (+ phases)
    ticks[16 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t_begin;
    ticks[16 * blockIdx.x + 1] = n_rot;
    for (int k = 0; k < 6; k++) ticks[16 * blockIdx.x + 2 + k] = tph[k];
    ticks[16 * blockIdx.x + 8] = t_begin - t_kernel;  // design matrix, AtA
  }
  const unsigned long long t_tail = ticks ? __builtin_amdgcn_s_memtime() : 0ull;
  // ---- smallest eigenvector -> E -> rank 2 (one lane per hypothesis)
  if (live && t == 0) {
    // only the FIRST element of the stable ascending eigenvalue sort (linalg.hpp:188-191) is
    // needed: the first strict minimum of the diagonal
    int col = 0;
    double wmin = L.A[0];
    for (int i = 1; i < 9; i++) {
      const double wi = L.A[i * 9 + i];
      if (wi < wmin) { wmin = wi; col = i; }
    }
    // conditioning of the choice: gap between the two smallest eigenvalues relative to the largest magnitude
    double w2 = 0.0, wabs = 0.0;
    bool have2 = false;
    for (int i = 0; i < 9; i++) {
      const double wi = L.A[i * 9 + i];
      wabs = fmax(wabs, fabs(wi));
      if (i != col && (!have2 || wi < w2)) { w2 = wi; have2 = true; }
    }
    double cond = (w2 - wmin) / wabs;
    for (int r = 0; r < 9; r++) L.E[r] = L.V[r * 9 + col];
    const double cond2 = rank2_project(L.E, L.A3, L.V3);
    if (!(cond > 0.0) || near_tie) cond = 0.0;  // also NaN
    cond = fmin(cond, cond2);
    for (int r = 0; r < 9; r++) E_out[(size_t)hyp * 9 + r] = L.E[r];
    cond_out[hyp] = cond;
  }
  if (ticks && lane == 0) ticks[16 * blockIdx.x + 9] = __builtin_amdgcn_s_memtime() - t_tail;  // eigenvector, rank-2 projection
}

// ------------------------------------------------------------------------------------------ scoring
// T:629-638 for one point and one E (E passed as 9 scalars so that it can live in SGPRs).
// Products by the homogeneous 1.0 are exact identities and are elided; the additions keep the
// reference's left-to-right order.
__device__ __forceinline__ double sampson_eval(double e0, double e1, double e2, double e3, double e4, double e5, double e6,
                                               double e7, double e8, double x, double y, double xp, double yp, double* den_out = nullptr) {
  const double Exx = e0 * x + e1 * y + e2;
  const double Exy = e3 * x + e4 * y + e5;
  const double Exz = e6 * x + e7 * y + e8;
  const double Etx = e0 * xp + e3 * yp + e6;
  const double Ety = e1 * xp + e4 * yp + e7;
  const double q = xp * Exx + yp * Exy + Exz;
  const double den = Exx * Exx + Exy * Exy + Etx * Etx + Ety * Ety + 1e-12;
  if (den_out) *den_out = den;
  return (q * q) / den;
}

#define SC_HB 8
#define SC_THREADS 256
// Distance between a device hypothesis and the reference's: |E_dev - E_ref|_max <= RANSAC_DEV_EPS / cond, cond = the conditioning
// estimate k_hypotheses reports (measured on MI355X over bench-like and fixture data: <= 3e-18 / cond, tools/ransac_cond_probe.py;
// the tests hold every fixture to 1e-17 / cond).  Exact host hypotheses carry cond = +inf.
#define RANSAC_DEV_EPS 1e-16
// counts[0][h] = #{err < thr}; counts[1][h] = #{err < thr (1 - rho)}; counts[2][h] = #{err < thr (1 + rho)} where rho bounds the
// relative change of a point's Sampson error when every entry of E moves by dE = RANSAC_DEV_EPS / cond[h]:
//   q = x'^T E x moves by <= dE S^2, S >= 1 + |x| + |y| of every point (both images);  near the threshold |q| = sqrt(thr den), so
//   q^2 moves by <= 2 dE S^2 / sqrt(thr den) relative;  den = |Ex|_xy^2 + |E^T x'|_xy^2 moves by <= 4 dE S / sqrt(den) relative;
//   rho = 2 (safety) * dE * (2 S^2 / sqrt(thr) + 4 S) / sqrt(den) = kappa[h] / sqrt(den),  kfac = 2 (2 S^2 / sqrt(thr) + 4 S).
// The reference's own count therefore lies in [counts[1], counts[2]].
__global__ __launch_bounds__(SC_THREADS) void k_score(const double* __restrict__ xi, const double* __restrict__ xj, int n,
                                                      const double* __restrict__ E, const double* __restrict__ cond, int H, double thr,
                                                      double kfac, int32_t* __restrict__ counts, double* __restrict__ cond_copy) {
  // counts / cond_copy may point into pinned HOST memory: the few posted writes per workgroup replace a DMA copy
  const int h0 = blockIdx.x * SC_HB;
  const int tid = threadIdx.x;
  // one packed counter per hypothesis: bits 0..19 mid, 20..39 lo, 40..59 hi (n < 2^20 is checked by the host)
  unsigned long long cnt[SC_HB];
  double kappa[SC_HB];
#pragma unroll
  for (int k = 0; k < SC_HB; k++) {
    cnt[k] = 0;
    const double c = cond[min(h0 + k, H - 1)];               // uniform
    kappa[k] = (c > 0.0 ? fmin(RANSAC_DEV_EPS / c, 1.0) : 1.0) * kfac;
  }
  const double2* __restrict__ pi = reinterpret_cast<const double2*>(xi);
  const double2* __restrict__ pj = reinterpret_cast<const double2*>(xj);
  for (int i = tid; i < n; i += SC_THREADS) {
    const double2 a = pi[i], b = pj[i];
#pragma unroll
    for (int k = 0; k < SC_HB; k++) {
      const int h = min(h0 + k, H - 1);          // uniform -> scalar loads of E
      const double* e = E + (size_t)h * 9;
      double den;
      const double err = sampson_eval(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8], a.x, a.y, b.x, b.y, &den);
      const double t = thr * fmin(kappa[k] * __builtin_amdgcn_rsq(den), 0.5);
      cnt[k] += ((err < thr) ? 1ull : 0ull) + ((err < thr - t) ? (1ull << 20) : 0ull) + ((err < thr + t) ? (1ull << 40) : 0ull);
    }
  }
  __shared__ unsigned long long part64[SC_THREADS / 64][SC_HB];
#pragma unroll
  for (int k = 0; k < SC_HB; k++) {
    unsigned long long v = cnt[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) part64[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < SC_HB && h0 + tid < H) {
    unsigned long long v = 0;
    for (int w = 0; w < SC_THREADS / 64; w++) v += part64[w][tid];
    counts[h0 + tid] = (int32_t)(v & 0xfffffull);
    counts[H + h0 + tid] = (int32_t)((v >> 20) & 0xfffffull);
    counts[2 * H + h0 + tid] = (int32_t)((v >> 40) & 0xfffffull);
    if (cond_copy) cond_copy[h0 + tid] = cond[h0 + tid];
  }
}

// rows of exact (host, libm) hypotheses replace the device's before scoring
__global__ void k_patch_E(double* __restrict__ E, double* __restrict__ cond, const int32_t* __restrict__ iters, const double* __restrict__ Ex,
                          int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m * 9) return;
  E[(size_t)iters[i / 9] * 9 + i % 9] = Ex[i];
  if (i % 9 == 0) cond[iters[i / 9]] = INFINITY;  // exact: no band
}

struct E9 { double e[9]; };
__global__ void k_sampson_mask(const double* __restrict__ xi, const double* __restrict__ xj, int n, E9 E, double thr, uint8_t* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double err = sampson_eval(E.e[0], E.e[1], E.e[2], E.e[3], E.e[4], E.e[5], E.e[6], E.e[7], E.e[8], xi[2 * i], xi[2 * i + 1], xj[2 * i],
                                  xj[2 * i + 1]);
  mask[i] = (err < thr) ? 1 : 0;
}

// csrc/hip/ransac_exact.cpp (g++, platform libm): eight_point_E of T:609-627 for the listed iterations
extern "C" void sfmx_exact_eight_point_batch(const double* xi, const double* xj, const int32_t* idx8, const int32_t* iters, int m, double* E_out);

// Relative half-width of the Sampson-error band around thr inside which a device hypothesis' verdict on a point is
// not trusted, and the conditioning below which a device hypothesis is not used at all.
static double env_double(const char* name, double dflt) {
  const char* e = getenv(name);
  return e ? atof(e) : dflt;
}
// Device hypotheses whose conditioning estimate is below this (incl. 0: nearly tied Jacobi pivot, NaN) are re-derived on the
// host: RANSAC_DEV_EPS / cond would be a distance of 1e-3 or more and the band of k_score too wide to be useful.
static double ransac_min_cond() { static const double v = env_double("SFMX_RANSAC_MIN_COND", 1e-13); return v; }
static inline size_t pad8(size_t v) { return (v + 7) & ~(size_t)7; }

extern "C" {

int sfmx_ransac_score_ex(sfmx_ctx* c, const double* xi, const double* xj, int n, const int32_t* idx8, int H, double thr,
                         int32_t* counts_out, int32_t* lo_out, int32_t* hi_out, uint8_t* flags_out, double* cond_out,
                         int32_t* best_iter, int32_t* best_count, double* E_out) {
  SFMX_REQUIRE(c, c && xi && xj && idx8 && n >= 8 && n < (1 << 20) && H > 0 && best_iter && best_count);
  const size_t pb = (size_t)n * 16, ib = (size_t)H * 32;
  // ---- octets with a repeated index (sampling is with replacement, T:665): their AtA has a null space of dimension
  // >= 2 and the reference's E is whatever its libm Jacobi lands on -- derived exactly on the host, while the device
  // builds the other hypotheses, and patched into the hypothesis buffer before scoring.
  std::vector<int32_t> exact_it;
  for (int h = 0; h < H; h++) {
    const int32_t* o = idx8 + (size_t)8 * h;
    bool rep = false;
    for (int a = 0; a < 8; a++) {
      if (o[a] < 0 || o[a] >= n) return sfmx_fail(c, SFMX_ERR_INVALID, "idx8 out of range", hipSuccess);
      for (int b = a + 1; b < 8; b++) rep |= (o[a] == o[b]);
    }
    if (rep) exact_it.push_back(h);
  }
  // one device slab [xi pb][xj pb][idx8 ib] filled by ONE transfer from the pinned staging slab; the staging slab
  // also holds the patch area [iters m*4, padded to 8][E m*72]
  const size_t in_bytes = 2 * pb + ib, patch_cap = (size_t)H * 84 + 32;  // + [cond m*8] of the second round
  const size_t cnt_bytes = pad8((size_t)H * 12);  // [mid H][lo H][hi H] int32, then [cond H] f64
  SFMX_HIP(c, c->d[0].ensure(in_bytes));
  SFMX_HIP(c, c->d[1].ensure(patch_cap));
  SFMX_HIP(c, c->d[3].ensure((size_t)H * 72));
  SFMX_HIP(c, c->d[4].ensure(cnt_bytes + (size_t)H * 8));
  SFMX_HIP(c, c->h[0].ensure(in_bytes + patch_cap));
  SFMX_HIP(c, c->h[1].ensure(cnt_bytes + (size_t)H * 8));
  char* hin = c->h[0].as<char>();
  memcpy(hin, xi, pb);
  memcpy(hin + pb, xj, pb);
  memcpy(hin + 2 * pb, idx8, ib);
  SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, hin, in_bytes, hipMemcpyHostToDevice, c->stream));
  c->resident_points = n;
  const double* d_xi = c->d[0].as<double>();
  const double* d_xj = reinterpret_cast<const double*>(c->d[0].as<char>() + pb);
  const int32_t* d_idx = reinterpret_cast<const int32_t*>(c->d[0].as<char>() + 2 * pb);
  double* d_cond = reinterpret_cast<double*>(c->d[4].as<char>() + cnt_bytes);
  double* d_E = c->d[3].as<double>();
  // S >= 1 + |x| + |y| over all points of both images (k_score's band, see there)
  double S = 1.0;
  for (int i = 0; i < n; i++) {
    S = fmax(S, 1.0 + fabs(xi[2 * i]) + fabs(xi[2 * i + 1]));
    S = fmax(S, 1.0 + fabs(xj[2 * i]) + fabs(xj[2 * i + 1]));
  }
  const double kfac = (S < 1e150 && thr > 0.0) ? 2.0 * (2.0 * S * S / sqrt(thr) + 4.0 * S) : INFINITY;
  KernelTimer t(c);
  t.start();
  // SFMX_RANSAC_TICKS=1: s_memtime ticks per Jacobi rotation on stderr, =2: per phase of a rotation (stamped build; diagnostic,
  // one extra synchronisation per call).  SFMX_RANSAC_HYP=legacy: the loop as of round 2 (A/B; identical results).
  static const int ticks_mode = getenv("SFMX_RANSAC_TICKS") ? atoi(getenv("SFMX_RANSAC_TICKS")) : 0;
  const bool hyp_legacy = getenv("SFMX_RANSAC_HYP") && strcmp(getenv("SFMX_RANSAC_HYP"), "legacy") == 0;
  const int hyp_blocks = (H + HPW - 1) / HPW;
  unsigned long long* d_ticks = nullptr;
  if (ticks_mode) SFMX_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_ticks), (size_t)hyp_blocks * 128));
#define HYP_ARGS d_xi, d_xj, n, d_idx, H, 120, d_E, d_cond, d_ticks
  if (ticks_mode == 2 && hyp_legacy) SFMX_PROF(c, KID_HYPOTHESES, (k_hypotheses<false, true><<<hyp_blocks, 64, 0, c->stream>>>(HYP_ARGS)));
  else if (ticks_mode == 2) SFMX_PROF(c, KID_HYPOTHESES, (k_hypotheses<true, true><<<hyp_blocks, 64, 0, c->stream>>>(HYP_ARGS)));
  else if (hyp_legacy) SFMX_PROF(c, KID_HYPOTHESES, (k_hypotheses<false><<<hyp_blocks, 64, 0, c->stream>>>(HYP_ARGS)));
  else SFMX_PROF(c, KID_HYPOTHESES, (k_hypotheses<true><<<hyp_blocks, 64, 0, c->stream>>>(HYP_ARGS)));
#undef HYP_ARGS
  SFMX_HIP(c, hipGetLastError());
  if (ticks_mode) {
    std::vector<unsigned long long> tk((size_t)hyp_blocks * 16);
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
    SFMX_HIP(c, hipMemcpy(tk.data(), d_ticks, tk.size() * 8, hipMemcpyDeviceToHost));
    SFMX_HIP(c, hipFree(d_ticks));
    double sum_t = 0, sum_r = 0, max_t = 0, ph[6] = {0, 0, 0, 0, 0, 0}, setup = 0, tail = 0, tail_max = 0;
    for (int b = 0; b < hyp_blocks; b++) {
      sum_t += (double)tk[16 * b];
      sum_r += (double)tk[16 * b + 1];
      max_t = fmax(max_t, (double)tk[16 * b]);
      for (int k = 0; k < 6; k++) ph[k] += (double)tk[16 * b + 2 + k];
      setup += (double)tk[16 * b + 8];
      tail += (double)tk[16 * b + 9];
      tail_max = fmax(tail_max, (double)tk[16 * b + 9]);
    }
    const double per = sum_r > 0 ? 1.0 / sum_r : 0.0;
    fprintf(stderr, "[sfmx] k_hypotheses<%s> H=%d: %.1f rotations per wave, %.0f ticks per rotation, slowest wave %.0f ticks", hyp_legacy ? "legacy" : "lean", H,
            sum_r / hyp_blocks, sum_t * per, max_t);
    fprintf(stderr, " | setup %.0f, tail %.0f (max %.0f) ticks per wave", setup / hyp_blocks, tail / hyp_blocks, tail_max);
    if (ticks_mode == 2)
      fprintf(stderr, " | own+keys %.0f | reduce+tests %.0f | lds reads %.0f | angle %.0f | rotate+stores %.0f | wait+barrier %.0f", ph[0] * per, ph[1] * per,
              ph[2] * per, ph[3] * per, ph[4] * per, ph[5] * per);
    fprintf(stderr, "\n");
  }
  // exact E of the listed iterations -> staging patch area -> d[1]; returns the device pointer of the m x 9 block
  char* hpatch = hin + in_bytes;
  auto upload_exact = [&](const std::vector<int32_t>& its, const double** d_rows) -> int {
    const int m = (int)its.size();
    const size_t ioff = pad8((size_t)m * 4);
    memcpy(hpatch, its.data(), (size_t)m * 4);
    sfmx_exact_eight_point_batch(xi, xj, idx8, its.data(), m, reinterpret_cast<double*>(hpatch + ioff));
    double* h_inf = reinterpret_cast<double*>(hpatch + ioff + (size_t)m * 72);  // cond = +inf for a compact second-round batch
    for (int k = 0; k < m; k++) h_inf[k] = INFINITY;
    SFMX_HIP(c, hipMemcpyAsync(c->d[1].p, hpatch, ioff + (size_t)m * 80, hipMemcpyHostToDevice, c->stream));
    *d_rows = reinterpret_cast<const double*>(c->d[1].as<char>() + ioff);
    return SFMX_OK;
  };
  if (!exact_it.empty()) {  // overlaps with k_hypotheses (the host part) and is ordered behind it (the scatter)
    // the scatter kernel reads the ~10 KB of iteration numbers and exact rows straight out of the pinned staging area
    const int m = (int)exact_it.size();
    const size_t ioff = pad8((size_t)m * 4);
    memcpy(hpatch, exact_it.data(), (size_t)m * 4);
    sfmx_exact_eight_point_batch(xi, xj, idx8, exact_it.data(), m, reinterpret_cast<double*>(hpatch + ioff));
    k_patch_E<<<(m * 9 + 255) / 256, 256, 0, c->stream>>>(d_E, d_cond, reinterpret_cast<const int32_t*>(hpatch),
                                                         reinterpret_cast<const double*>(hpatch + ioff), m);
  }
  // counts and a copy of the conditioning estimates go straight into the pinned result slab (zero-copy writes)
  int32_t* h_cnt = c->h[1].as<int32_t>();
  double* h_cond = reinterpret_cast<double*>(c->h[1].as<char>() + cnt_bytes);
  SFMX_PROF(c, KID_SCORE, (k_score<<<(H + SC_HB - 1) / SC_HB, SC_THREADS, 0, c->stream>>>(d_xi, d_xj, n, d_E, d_cond, H, thr, kfac, h_cnt, h_cond)));
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  t.collect();
  std::vector<uint8_t> exact((size_t)H, 0);
  for (int32_t h : exact_it) exact[(size_t)h] = 1;
  // ---- ill-conditioned device hypotheses (eigenvalue gap, rank-2 gap, nearly tied pivot): second, rare round --
  // exact E on the host, scored as a compact batch, rows and counts patched in
  std::vector<int32_t> redo;
  const double min_cond = ransac_min_cond();
  for (int h = 0; h < H; h++)
    if (!exact[(size_t)h] && !(h_cond[h] >= min_cond)) redo.push_back(h);
  if (!redo.empty()) {
    const int m = (int)redo.size();
    const double* d_rows = nullptr;
    const int rc = upload_exact(redo, &d_rows);
    if (rc != SFMX_OK) return rc;
    SFMX_HIP(c, c->h[2].ensure((size_t)m * 12 + 64));
    int32_t* c2 = c->h[2].as<int32_t>();
    k_patch_E<<<(m * 9 + 255) / 256, 256, 0, c->stream>>>(d_E, d_cond, c->d[1].as<int32_t>(), d_rows, m);
    k_score<<<(m + SC_HB - 1) / SC_HB, SC_THREADS, 0, c->stream>>>(d_xi, d_xj, n, d_rows, d_rows + (size_t)9 * m, m, thr, kfac, c2, nullptr);
    SFMX_HIP(c, hipGetLastError());
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < m; k++) {
      h_cnt[redo[(size_t)k]] = c2[k];
      exact[(size_t)redo[(size_t)k]] = 1;
    }
  }
  // exact hypotheses: the count is the reference's (the Sampson arithmetic is bit-exact for a given E)
  for (int h = 0; h < H; h++)
    if (exact[(size_t)h]) h_cnt[H + h] = h_cnt[2 * H + h] = h_cnt[h];
  if (E_out) {
    SFMX_HIP(c, hipMemcpyAsync(E_out, d_E, (size_t)H * 72, hipMemcpyDeviceToHost, c->stream));
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (counts_out) memcpy(counts_out, h_cnt, (size_t)H * 4);
  if (lo_out) memcpy(lo_out, h_cnt + H, (size_t)H * 4);
  if (hi_out) memcpy(hi_out, h_cnt + 2 * H, (size_t)H * 4);
  if (flags_out) memcpy(flags_out, exact.data(), (size_t)H);
  if (cond_out)
    for (int h = 0; h < H; h++) cond_out[h] = exact[(size_t)h] ? INFINITY : h_cond[h];
  // argmax with the LOWEST iteration on ties (the reference's strict '>' at T:673)
  int32_t bi = 0, bc = h_cnt[0];
  for (int h = 1; h < H; h++)
    if (h_cnt[h] > bc) { bc = h_cnt[h]; bi = h; }
  *best_iter = bi;
  *best_count = bc;
  return SFMX_OK;
}

int sfmx_ransac_score(sfmx_ctx* c, const double* xi, const double* xj, int n, const int32_t* idx8, int H, double thr,
                      int32_t* counts_out, int32_t* best_iter, int32_t* best_count, double* E_out) {
  return sfmx_ransac_score_ex(c, xi, xj, n, idx8, H, thr, counts_out, nullptr, nullptr, nullptr, nullptr, best_iter, best_count, E_out);
}

// xi == xj == NULL reuses the correspondences left in HBM by the preceding sfmx_ransac_score call
// (n must match); otherwise they are uploaded (n*32 bytes).
int sfmx_sampson_mask(sfmx_ctx* c, const double* xi, const double* xj, int n, const double* E9in, double thr, uint8_t* mask_out,
                      int32_t* count_out) {
  SFMX_REQUIRE(c, c && E9in && mask_out && n > 0 && ((xi && xj) || (!xi && !xj && c->resident_points == n)));
  const size_t pb = (size_t)n * 16;
  SFMX_HIP(c, c->h[2].ensure((size_t)n + 64));
  if (xi) {  // same layout as sfmx_ransac_score leaves behind: [xi pb][xj pb]
    SFMX_HIP(c, c->d[0].ensure(2 * pb));
    SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, xi, pb, hipMemcpyHostToDevice, c->stream));
    SFMX_HIP(c, hipMemcpyAsync(c->d[0].as<char>() + pb, xj, pb, hipMemcpyHostToDevice, c->stream));
    c->resident_points = n;
  }
  const double* d_xi = c->d[0].as<double>();
  const double* d_xj = reinterpret_cast<const double*>(c->d[0].as<char>() + pb);
  E9 E;
  memcpy(E.e, E9in, 72);
  k_sampson_mask<<<(n + 255) / 256, 256, 0, c->stream>>>(d_xi, d_xj, n, E, thr, c->h[2].as<uint8_t>());  // written into pinned memory
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(mask_out, c->h[2].p, (size_t)n);
  int32_t cnt = 0;
  for (int i = 0; i < n; i++) cnt += mask_out[i];  // n <= a few thousand bytes
  if (count_out) *count_out = cnt;
  return SFMX_OK;
}

}  // extern "C"
