// klt.hip — pyramidal LK forward + backward tracking with the forward-backward test, for gfx950.
//
// Replaces KLTTracker::track_one / lk_step and the FB test of KLTTracker::step
// (reference cpp/src/templering_sfm.cpp T:356-362, 402-460).
//
// Mapping: ONE 64-lane wavefront per track (block = 1 wave, so __syncthreads() is a wave-local
// ordering point and every wave may run its own trip count: the early exit at T:416 is per track).
//
//  * LDS staging: a 32x32 window of both pyramid images around the current estimate is copied to
//    LDS as f32 (u8 -> f32 is exact) once per level and re-staged only when the (2r+5)-wide
//    footprint of an lk_step leaves it.
//  * sample grid (parallel).  sample_bilinear (T:183-198) is separable in what it needs of a coordinate:
//    (floor, fraction, in-image test).  An lk_step evaluates I1 at (xx+1,yy), (xx-1,yy), (xx,yy+1), (xx,yy-1),
//    (xx,yy) and I0 at (xx,yy) for xx = x+dx, yy = y+dy (T:436-441): 6 samples x (2r+1)^2 pixels, but only
//    ~(2r+3)^2 + (2r+1)^2 DISTINCT ones.  Per step the wave builds one descriptor per distinct coordinate
//    -- the reference's own expressions fl(x+dx), fl(fl(x+dx)+1), fl(fl(x+dx)-1), matched BITWISE against the
//    neighbouring fl(x+(dx+-1)) (they differ in the last bit where x+dx crosses a power of two; such values get
//    descriptor slots of their own, so the grid is exact by construction) -- then each distinct sample once
//    (rows first: v00*(1-fx)+v10*fx, then v0*(1-fy)+v1*fy, the reference's order), and the pixels read them.
//    A step with more extra coordinates than the descriptor table holds takes the per-pixel path.
//  * ordered reduction (serial by contract): FP64 addition is not associative and parity is
//    bit-exact, so the (2r+1)^2 products of each accumulator are added in the reference's
//    (dy outer, dx inner) sequence -- by default as chains of v_mfma_f64_4x4x4 with B = 1.0 (one matrix instruction = four
//    fused multiply-adds in ascending k, each rounded to FP64: four of the reference's additions for 16 accumulators; measured,
//    tools/probes/mfma_f64_order.hip), or as v_add_f64 chains in lanes 0..4 (SFMX_KLT_SUMS=valu, and for radius 1).
//  * 2x2 solve (the three divisions of T:453-455 in three lanes of one instruction), hypot-based stop test
//    (glibc-compatible hypot, sfmx_math.h), level loop, then the backward pass from the forward result and
//    keep = !(hypot(back - p0) >= fb_thresh).
//
// No FMA contraction anywhere (-ffp-contract=off); FP64 division and sqrt are the correctly
// rounded forms.
#include "sfmx_internal.h"

#include <cstdlib>
#include <vector>

#define KLT_P 32          // staged window is KLT_P x KLT_P pixels
// LDS row stride (floats) of the staged windows.  Lane l owns window pixel l = (dy+r)*(2r+1)+(dx+r)
// and reads win[(y0+dy)*PS + x0+dx]; with PS = 32 + (2r+1) the bank (address mod 32) of that read is
// l + const, so 32 consecutive lanes hit 32 distinct banks.  (Stride 33 put every anti-diagonal of
// the 11x11 window on one bank: rocprofv3 showed 2.6 conflict cycles per LDS instruction.)
#define KLT_PS_FOR(r) (32 + 2 * (r) + 1)
#define KLT_PS_MAX KLT_PS_FOR(KLT_MAX_R)
#define KLT_MAX_R 7
#define KLT_MAX_NPIX ((2 * KLT_MAX_R + 1) * (2 * KLT_MAX_R + 1))

struct Tap {   // one coordinate of a bilinear sample: clamped window index, fraction, in-image mask
  int l;       // index inside the staged window (clamped to [0, KLT_P-2])
  double f;    // fractional part (0 when the tap is outside the image, so every lerp stays finite)
  double m;    // 1.0 inside the image, 0.0 outside
};
__device__ __forceinline__ Tap make_tap(double v, int extent, int origin) {
  Tap t;
  const int i0 = sfmx::floor_to_int_x86(v);
  const bool ok = (i0 >= 0) && (i0 < extent - 1);  // x0 >= 0 && x0+1 < w   (T:188)
  t.f = ok ? v - (double)i0 : 0.0;
  t.m = ok ? 1.0 : 0.0;
  t.l = min(max(i0 - origin, 0), KLT_P - 2);       // inside the staged window by construction when ok
  return t;
}
// T:183-198 on the staged window, branch-free: the four pixels are always read (clamped address) and the
// in-image test multiplies the result by 1.0 or 0.0 -- exact (v is finite and >= 0), and it keeps the
// loads unconditional so that the reads of all six samples of a pixel are issued back to back (a
// `cond ? v : 0` select is turned back into a branch around the loads by the compiler).
template <int KLT_PS>
__device__ __forceinline__ double sample_lds(const float* __restrict__ win, const Tap& cx, const Tap& cy) {
  const float* p = win + cy.l * KLT_PS + cx.l;
  const double v00 = (double)p[0], v10 = (double)p[1], v01 = (double)p[KLT_PS], v11 = (double)p[KLT_PS + 1];
  const double v0 = v00 * (1 - cx.f) + v10 * cx.f;
  const double v1 = v01 * (1 - cx.f) + v11 * cx.f;
  const double v = v0 * (1 - cy.f) + v1 * cy.f;
  return v * (cx.m * cy.m);
}

// Both windows are staged together: every global load of a lane is issued before the first LDS store, so one staging
// costs about one L2 round trip.  A window that lies inside the image (the usual case; wave-uniform test) is fetched as
// 16 bytes per lane and image -- lane = (row, half row), four byte-aligned dword loads -- instead of 16 byte loads.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
template <int KLT_PS>
__device__ __forceinline__ void stage_windows(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int w, int h, int ox,
                                              int oy, float* __restrict__ win0, float* __restrict__ win1, int lane) {
  if (ox >= 0 && oy >= 0 && ox + KLT_P <= w && oy + KLT_P <= h) {
    static_assert(KLT_P == 32, "two lanes per window row");
    const int row = lane >> 1, c0 = (lane & 1) * 16;
    const size_t off = (size_t)(oy + row) * w + ox + c0;
    const u32_unaligned* a = reinterpret_cast<const u32_unaligned*>(img0 + off);
    const u32_unaligned* b = reinterpret_cast<const u32_unaligned*>(img1 + off);
    uint32_t va[4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { va[k] = a[k]; vb[k] = b[k]; }
    float* d0 = win0 + row * KLT_PS + c0;
    float* d1 = win1 + row * KLT_PS + c0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        d0[4 * k + q] = (float)((va[k] >> (8 * q)) & 0xffu);
        d1[4 * k + q] = (float)((vb[k] >> (8 * q)) & 0xffu);
      }
    }
    return;
  }
  constexpr int N = KLT_P * KLT_P / 64;  // 16 pixels per lane per image
  const int px = lane % KLT_P, py0 = lane / KLT_P;  // lane covers column px of rows py0, py0+2, ...
  const int gx = ox + px;
  const bool xok = gx >= 0 && gx < w;
  uint8_t a[N], b[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int gy = oy + py0 + 2 * k;
    const bool ok = xok && gy >= 0 && gy < h;
    const size_t off = ok ? (size_t)gy * w + gx : 0;
    a[k] = img0[off];
    b[k] = img1[off];
    if (!ok) { a[k] = 0; b[k] = 0; }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int o = (py0 + 2 * k) * KLT_PS + px;
    win0[o] = (float)a[k];
    win1[o] = (float)b[k];
  }
}

// wave-uniform broadcast of a double held by `src` lane (v_readlane: no LDS round trip)
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// hypot(sx,sy) < lim  with glibc's hypot semantics.  hypot is within a few ulp of sqrt(sx^2+sy^2), so
// outside a 1e-9 relative band around lim^2 the squared comparison decides; inside it the exact
// restatement is evaluated.
__device__ __forceinline__ bool hypot_below(double sx, double sy, double lim) {
  const double q = sx * sx + sy * sy, l2 = lim * lim;
  if (q < l2 * (1.0 - 1e-9)) return true;
  if (q > l2 * (1.0 + 1e-9)) return false;  // also taken for inf; NaN falls through to the exact form
  return sfmx::hypot_glibc(sx, sy) < lim;
}

// integer part of a coordinate for window bookkeeping only (saturating; NaN -> far away)
__device__ __forceinline__ int book_floor(double v) {
  if (!(v > -1.0e9 && v < 1.0e9)) return (int)0x40000000;
  return (int)floor(v);
}

// LDS plan of one track (one wavefront) for window radius r.  MS (ordered sums on the FP64 matrix core): the product rows are
// padded to a multiple of four entries and a sixth row holds zeros (operand lanes without an accumulator read it).
template <int r, bool MS = false>
struct KltLds {
  static constexpr int side = 2 * r + 1, npix = side * side, npad = MS ? ((npix + 3) & ~3) : ((npix + 1) & ~1);
  static constexpr int prows = MS ? 6 : 5;
  static constexpr int PS = KLT_PS_FOR(r);
  static constexpr int NCAN = 2 * r + 3;        // grid coordinates per axis: slot 0 = fl(fl(x-r)-1), slot s = fl(x+s-1-r), slot 2r+2 = fl(fl(x+r)+1)
  static constexpr int GS = NCAN + 1;           // grid row stride (doubles)
  static constexpr int win_floats = (2 * KLT_P * PS + 3) & ~3;
  static constexpr size_t o_g1 = (size_t)win_floats * 4;         // double G1[GS][GS]: I1 on the grid
  static constexpr size_t o_g0 = o_g1 + (size_t)GS * GS * 8;     // double G0[GS][GS]: I0 on the grid
  static constexpr size_t o_prod = o_g0 + (size_t)GS * GS * 8;   // double prod[5][npad]
  static constexpr size_t o_tapf = o_prod + (size_t)prows * npad * 8; // double2 {f, 1-f} ({0, 0} outside the image) [2 axes][32]: taps of the grid slots
  static constexpr size_t o_tapi = o_tapf + (size_t)2 * 32 * 16;  // int byte offset of the tap's column / row inside a window, [2][32]
  static constexpr size_t o_xtapf = o_tapi + (size_t)2 * 32 * 4;   // double2 taps of off-grid neighbours fl(c_d + 1) / fl(c_d - 1): [2 kinds][2 axes][16]
  static constexpr size_t o_xtapi = o_xtapf + (size_t)2 * 2 * 16 * 16;  // their byte offsets
  static constexpr size_t bytes = o_xtapi + (size_t)2 * 2 * 16 * 4;
  static constexpr int NS = NCAN * NCAN + side * side;           // samples per step: I1 on the grid, I0 on its centre
  static constexpr int SR = (NS + 63) / 64;                      // sample rounds
};

template <int CTRL>
__device__ __forceinline__ long long dpp_i64(long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(v & 0xffffffffll), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xF, 0xF, false);
  return ((long long)hi << 32) | (unsigned int)lo;
}

// coordinate of grid slot s on an axis whose step coordinate is v, with the reference's expressions (T:436-439):
// xx = v + (double)d, xx - 1 for the first slot, xx + 1 for the last one
template <int r>
__device__ __forceinline__ double slot_coord(double v, int s) {
  const int d = min(max(s - 1, 0), 2 * r) - r;
  const double ca = v + (double)d;
  return s == 0 ? ca - 1 : (s == 2 * r + 2 ? ca + 1 : ca);
}

// T:183-198 for one (column tap, row tap) pair on a staged window; omx/omy = 1 - f of the taps
template <int KLT_PS>
__device__ __forceinline__ double lerp_lds(const float* __restrict__ win, const Tap& cx, double omx, const Tap& cy, double omy) {
  const float* p = win + cy.l * KLT_PS + cx.l;
  const double v00 = (double)p[0], v10 = (double)p[1], v01 = (double)p[KLT_PS], v11 = (double)p[KLT_PS + 1];
  const double v0 = v00 * omx + v10 * cx.f;
  const double v1 = v01 * omx + v11 * cx.f;
  return (v0 * omy + v1 * cy.f) * (cx.m * cy.m);
}

// STAMP: diagnostic build (SFMX_KLT_STAMPS=1) that accumulates s_memtime deltas per phase of the step loop for track 0
// into stamps[0..7] (a buffer nothing else reads); the production instantiation has no stamp code.
// PIPE (window radius 4 and 5: 64 < npix <= 128): the ordered sums over the first 64 pixels run UNDER the second half of the sample
// grid and of the products -- see the PIPE branch below.  Same arithmetic, another instruction order.
// MSUM: the ordered sums as chains of v_mfma_f64_4x4x4 with B = 1.0 -- see the MSUM branch below.
template <int r, bool STAMP, bool PIPE = false, bool MSUM = false>
__global__ __launch_bounds__(64) void k_klt_track(PyrDesc A, PyrDesc B, const double* __restrict__ xy_in, int n, int levels,
                                                  int iters, double fb_thresh, double* __restrict__ xy_fwd,
                                                  double* __restrict__ xy_back, uint8_t* __restrict__ keep,
                                                  unsigned long long* __restrict__ step_counter, unsigned long long* __restrict__ stamps,
                                                  int wave_prio) {
  // wave priority (s_setprio, 0..3): above the bulk kernels of the other lanes, below the BA chain (3)
  if (wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  using L = KltLds<r, MSUM>;
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int k) {
    if (STAMP) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tph[k] += now - tlast;
      tlast = now;
    }
  };
  if (STAMP) tlast = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) unsigned char smem[];
  float* win0 = reinterpret_cast<float*>(smem);                 // template image window (I0 of lk_step)
  float* win1 = win0 + KLT_P * KLT_PS_FOR(r);                   // current image window  (I1 of lk_step)
  double* G1 = reinterpret_cast<double*>(smem + L::o_g1);
  double* G0 = reinterpret_cast<double*>(smem + L::o_g0);
  double* prod = reinterpret_cast<double*>(smem + L::o_prod);
  double2* tapf = reinterpret_cast<double2*>(smem + L::o_tapf);
  int* tapo = reinterpret_cast<int*>(smem + L::o_tapi);
  double2* xtapf = reinterpret_cast<double2*>(smem + L::o_xtapf);
  int* xtapo = reinterpret_cast<int*>(smem + L::o_xtapi);
  const int lane = threadIdx.x;
  const int track = blockIdx.x;
  if (track >= n) return;
  constexpr int side = L::side, npix = L::npix, npad = L::npad, NCAN = L::NCAN, GS = L::GS;
  constexpr int KLT_PS = KLT_PS_FOR(r);
  constexpr int PP = (npix + 63) / 64;      // pixels per lane
  static_assert(NCAN <= 32, "one descriptor lane per grid slot and axis");
  // the samples of this lane, fixed for the whole kernel.  Sample e < NCAN^2 is I1 at grid point (e / NCAN, e % NCAN), the
  // others are I0 at the centre points (slots 1..2r+1); a lane without a sample in the last round recomputes sample 0.
  // PIPE: the samples are ordered so that the first PR1 rounds hold everything the products of pixels 0..63 read (I1 rows
  // 0..RH1 of the grid, I0 under those pixels) and the remaining PR2 rounds the rest
  constexpr int RH1 = 63 / side + 2;                                                  // last I1 grid row pixels 0..63 touch
  constexpr int PN1 = (RH1 + 1) * NCAN + 64, PR1 = (PN1 + 63) / 64;
  constexpr int PN2 = (NCAN - RH1 - 1) * NCAN + (npix > 64 ? npix - 64 : 0), PR2 = (PN2 + 63) / 64;
  constexpr int NSR = PIPE ? PR1 + PR2 : L::SR;                                       // sample rounds
  int samp_cd[NSR], samp_rd[NSR];
  bool samp_i0[NSR], samp_ok[NSR];
  double* samp_out[NSR];
#pragma unroll
  for (int k = 0; k < NSR; k++) {
    if constexpr (PIPE) {
      const bool first = k < PR1;
      const int idx = lane + 64 * (first ? k : k - PR1);
      const int n_i1 = first ? (RH1 + 1) * NCAN : (NCAN - RH1 - 1) * NCAN;
      const int n_all = first ? PN1 : PN2;
      samp_ok[k] = idx < n_all;
      const int id = samp_ok[k] ? idx : 0;
      samp_i0[k] = id >= n_i1;
      const int pix = (first ? 0 : 64) + (id - n_i1);                                  // I0: the pixel it lies under
      samp_cd[k] = samp_i0[k] ? pix % side + 1 : id % NCAN;
      samp_rd[k] = samp_i0[k] ? pix / side + 1 : (first ? 0 : RH1 + 1) + id / NCAN;
    } else {
      int e = lane + 64 * k;
      samp_ok[k] = e < L::NS;
      if (e >= L::NS) e = 0;
      samp_i0[k] = e >= NCAN * NCAN;
      samp_cd[k] = samp_i0[k] ? (e - NCAN * NCAN) % side + 1 : e % NCAN;
      samp_rd[k] = samp_i0[k] ? (e - NCAN * NCAN) / side + 1 : e / NCAN;
    }
    samp_out[k] = (samp_i0[k] ? G0 : G1) + samp_rd[k] * GS + samp_cd[k];
  }
  const double p0x = xy_in[2 * track], p0y = xy_in[2 * track + 1];
  double px = p0x, py = p0y;
  unsigned int steps = 0, slow_steps = 0;
  if constexpr (MSUM) {  // the zero row and the pad entries of the product rows: written once, never by the products
    for (int i = lane; i < npad; i += 64) prod[5 * npad + i] = 0.0;
    constexpr int NPADS = npad - npix;  // 3 for every odd window side
    if (NPADS > 0 && lane < 5 * NPADS) prod[(lane / (NPADS > 0 ? NPADS : 1)) * npad + npix + lane % (NPADS > 0 ? NPADS : 1)] = 0.0;
    __syncthreads();
  }
  // lane roles of the coordinate check: lanes 0..31 the x axis, 32..63 the y axis, lane (axis, dt) owns offset d = dt - r
  const int dt = lane & 31, daxis = lane >> 5;

  for (int dir = 0; dir < 2; ++dir) {
    for (int l = levels - 1; l >= 0; --l) {
      const uint8_t* img0 = dir == 0 ? A.px[l] : B.px[l];
      const uint8_t* img1 = dir == 0 ? B.px[l] : A.px[l];
      const int w = A.w[l], h = A.h[l];
      const double scale = 1.0 / (double)(1 << l);
      const double plx = px * scale, ply = py * scale;
      double dlx = 0.0, dly = 0.0;
      int ox = (int)0x7fffff00, oy = (int)0x7fffff00;  // no window staged yet
      for (int it = 0; it < iters; ++it) {
        const double x = plx + dlx, y = ply + dly;
        stamp(5);  // level / loop bookkeeping
        // ---- make sure the staged window covers the footprint [b-r-2, b+r+3] of this step
        const int bx = book_floor(x), by = book_floor(y);
        const bool touches = (bx + r + 3 >= 0) && (bx - r - 2 < w) && (by + r + 3 >= 0) && (by - r - 2 < h);
        if (!touches) {
          // Every sample of this step is 0.0 in the reference (T:188) => A = 0, detA = 0 => step {0,0} (T:452)
          // => hypot(0,0) < 1e-3 ends the level (T:416).  Nothing to read: the window may be unstaged, and
          // stale LDS bits must never reach the mask-multiply of a sample (NaN * 0 = NaN).
          ++steps;
          break;
        }
        {
          const bool covered = (bx - r - 2 >= ox) && (bx + r + 3 < ox + KLT_P) && (by - r - 2 >= oy) && (by + r + 3 < oy + KLT_P);
          if (!covered) {
            ox = bx - (KLT_P / 2 - 1);
            oy = by - (KLT_P / 2 - 1);
            __syncthreads();
            stage_windows<KLT_PS>(img0, img1, w, h, ox, oy, win0, win1, lane);
            __syncthreads();
          }
        }
        stamp(0);  // window staging
        // ---- coordinate taps of the grid slots, one lane per slot (lanes 0.. the x axis, 32.. the y axis), and: which of the
        // reference's neighbour coordinates are ON the grid?  Per axis the reference evaluates c_d = fl(v + d), fl(c_d + 1),
        // fl(c_d - 1) for d = -r..r (T:436-439).  Slot s = d+r+1 holds c_d, slot 0 holds fl(c_{-r} - 1), slot 2r+2 holds
        // fl(c_r + 1); fl(c_d + 1) IS slot s+1 iff it equals that slot's coordinate bitwise (it differs in the last bit
        // where v + d crosses a power of two), likewise fl(c_d - 1) and slot s-1.
        // mis_p / mis_m: bit d+r (x axis) and 32+d+r (y axis) set = that neighbour needs a sample of its own.
        unsigned long long mis_p, mis_m;
        {
          const double cv = slot_coord<r>(daxis ? y : x, min(dt, NCAN - 1));
          const long long cbits = __double_as_longlong(cv);
          // the cross-lane reads run with every lane enabled (a DPP read of a lane that EXEC masks off returns the fill value)
          long long next_bits, prev_bits;
          if constexpr (NCAN <= 16) {  // all slots of an axis sit in one DPP row
            next_bits = dpp_i64<0x101>(cbits);  // row_shl:1 = lane+1
            prev_bits = dpp_i64<0x111>(cbits);  // row_shr:1 = lane-1
          } else {
            next_bits = __shfl_down(cbits, 1, 64);
            prev_bits = __shfl_up(cbits, 1, 64);
          }
          const bool inner = dt >= 1 && dt <= side;  // slots of c_d
          const bool bad_p = inner & (__double_as_longlong(cv + 1) != next_bits);
          const bool bad_m = inner & (__double_as_longlong(cv - 1) != prev_bits);
          mis_p = __ballot(bad_p) >> 1;  // slot s -> bit d+r = s-1
          mis_m = __ballot(bad_m) >> 1;
          if (dt < NCAN) {
            // An out-of-image tap gets BOTH weights 0: every lerp it takes part in is then +0.0, which is what the
            // reference returns for such a sample (T:188; pixel values are finite and >= 0), with no mask arithmetic.
            const Tap tp = make_tap(cv, daxis ? h : w, daxis ? oy : ox);
            const bool ok = tp.m != 0.0;
            tapf[daxis * 32 + dt] = make_double2(tp.f, ok ? 1 - tp.f : 0.0);
            tapo[daxis * 32 + dt] = tp.l * (daxis ? KLT_PS * 4 : 4);  // byte offset of the tap's row / column in a window
          }
          if (bad_p | bad_m) {  // off-grid neighbours get taps of their own (rare: skipped when no lane needs one)
            const double ce = bad_p ? cv + 1 : cv - 1;  // a slot with BOTH neighbours off the grid writes the second one below
            const Tap tp = make_tap(ce, daxis ? h : w, daxis ? oy : ox);
            const int e = (bad_p ? 0 : 32) + daxis * 16 + (dt - 1);
            xtapf[e] = make_double2(tp.f, tp.m != 0.0 ? 1 - tp.f : 0.0);
            xtapo[e] = tp.l * (daxis ? KLT_PS * 4 : 4);
            if (bad_p & bad_m) {
              const Tap tq = make_tap(cv - 1, daxis ? h : w, daxis ? oy : ox);
              xtapf[32 + daxis * 16 + (dt - 1)] = make_double2(tq.f, tq.m != 0.0 ? 1 - tq.f : 0.0);
              xtapo[32 + daxis * 16 + (dt - 1)] = tq.l * (daxis ? KLT_PS * 4 : 4);
            }
          }
        }
        __syncthreads();
        double acc = 0.0;
        if constexpr (PIPE) {
          static_assert(npix > 64 && npix <= 128, "two product rounds");
          // the pieces of a step as callable blocks (everything is inlined; the indices are compile-time constants)
          double2 fx[NSR], fy[NSR];
          int po[NSR];
          float pv[NSR][4];
          auto taps_of = [&](int k) {
            fx[k] = tapf[samp_cd[k]];
            fy[k] = tapf[32 + samp_rd[k]];
            po[k] = tapo[samp_cd[k]] + tapo[32 + samp_rd[k]];
          };
          auto pixels_of = [&](int k) {
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(samp_i0[k] ? win0 : win1) + po[k]);
            pv[k][0] = p[0]; pv[k][1] = p[1]; pv[k][2] = p[KLT_PS]; pv[k][3] = p[KLT_PS + 1];
          };
          auto lerp_store = [&](int k) {
            const double v0 = (double)pv[k][0] * fx[k].y + (double)pv[k][1] * fx[k].x;
            const double v1 = (double)pv[k][2] * fx[k].y + (double)pv[k][3] * fx[k].x;
            const double val = v0 * fy[k].y + v1 * fy[k].x;
            if (samp_ok[k]) samp_out[k][0] = val;
          };
          const bool any_mis = (mis_p | mis_m) != 0ull;  // wave-uniform
          if (any_mis) ++slow_steps;
          double gxp, gxm, gyp, gym, gcc, irf;
          auto product_reads = [&](int q) {
            const int pix = lane + 64 * q < npix ? lane + 64 * q : 0;
            const int i = pix / side, j = pix % side;
            const double* gc1 = G1 + (i + 1) * GS + (j + 1);
            gxp = gc1[1]; gxm = gc1[-1]; gyp = gc1[GS]; gym = gc1[-GS]; gcc = gc1[0];
            irf = G0[(i + 1) * GS + (j + 1)];
            if (any_mis) {
              auto lerp = [&](const double2& fxx, const double2& fyy, int off) {
                const float* p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(win1) + off);
                const double v0 = (double)p[0] * fxx.y + (double)p[1] * fxx.x;
                const double v1 = (double)p[KLT_PS] * fxx.y + (double)p[KLT_PS + 1] * fxx.x;
                return v0 * fyy.y + v1 * fyy.x;
              };
              if ((mis_p >> j) & 1ull) gxp = lerp(xtapf[j], tapf[32 + i + 1], xtapo[j] + tapo[32 + i + 1]);
              if ((mis_m >> j) & 1ull) gxm = lerp(xtapf[32 + j], tapf[32 + i + 1], xtapo[32 + j] + tapo[32 + i + 1]);
              if ((mis_p >> (32 + i)) & 1ull) gyp = lerp(tapf[j + 1], xtapf[16 + i], tapo[j + 1] + xtapo[16 + i]);
              if ((mis_m >> (32 + i)) & 1ull) gym = lerp(tapf[j + 1], xtapf[48 + i], tapo[j + 1] + xtapo[48 + i]);
            }
          };
          auto product_stores = [&](int q) {
            const int pix = lane + 64 * q;
            if (pix < npix) {
              const double Ix = 0.5 * (gxp - gxm), Iy = 0.5 * (gyp - gym), err = irf - gcc;
              prod[0 * npad + pix] = Ix * Ix;
              prod[1 * npad + pix] = Ix * Iy;
              prod[2 * npad + pix] = Iy * Iy;
              prod[3 * npad + pix] = Ix * err;
              prod[4 * npad + pix] = Iy * err;
            }
          };
          // VALU sums: one accumulator per lane 0..4; the lanes above repeat lane 4's chain, so that no EXEC change separates the
          // adds from the instructions they are meant to be scheduled between.  Matrix-core sums (MSUM, see the plain schedule
          // below for the operand layout): the chain instructions run in their own pipe, next to the VALU work between them.
          // A chain unit is a pair of addends (VALU) or a group of four (MSUM); the first 64 pixels are 4 * CU units.
          constexpr int CU = MSUM ? 4 : 8, CTOT = MSUM ? npad / 4 : npix / 2;
          const double2* cq = reinterpret_cast<const double2*>(prod + (lane < 5 ? lane : 4) * npad);
          const double* mq = prod + ((lane & 15) < 5 ? (lane & 15) : 5) * npad + (lane >> 4);
          auto chain = [&](int i0, int cnt) {  // units i0 .. i0 + cnt - 1 of the accumulator's products, in order
            if constexpr (MSUM) {
              double v[8];
#pragma unroll
              for (int k = 0; k < 8; k++)
                if (k < cnt) v[k] = mq[4 * (i0 + k)];
#pragma unroll
              for (int k = 0; k < 8; k++)
                if (k < cnt) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(v[k], 1.0, acc, 0, 0, 0);
            } else {
              double2 v[8];
#pragma unroll
              for (int k = 0; k < 8; k++)
                if (k < cnt) v[k] = cq[i0 + k];
#pragma unroll
              for (int k = 0; k < 8; k++)
                if (k < cnt) { acc += v[k].x; acc += v[k].y; }
            }
          };
          // ---- first half: grid rounds 0..PR1-1, products of pixels 0..63
#pragma unroll
          for (int k = 0; k < PR1; k++) taps_of(k);
#pragma unroll
          for (int k = 0; k < PR1; k++) pixels_of(k);
#pragma unroll
          for (int k = 0; k < PR1; k++) lerp_store(k);
          __syncthreads();
          stamp(1);
          product_reads(0);
          product_stores(0);
          __syncthreads();
          stamp(2);
          // ---- second half of the grid and of the products, with the 64 ordered adds over pixels 0..63 in between: the adds are a
          // chain of dependent instructions (~10 cycles each) that leaves the issue slots free for everything else
#pragma unroll
          for (int k = PR1; k < NSR; k++) taps_of(k);
          chain(0, CU);
#pragma unroll
          for (int k = PR1; k < NSR; k++) pixels_of(k);
          chain(CU, CU);
#pragma unroll
          for (int k = PR1; k < NSR; k++) lerp_store(k);
          chain(2 * CU, CU);
          product_reads(1);  // after the stores of the grid's second half (same wave: LDS operations keep their order)
          chain(3 * CU, CU);
          product_stores(1);
          __syncthreads();
          // ---- the rest of the ordered sums: pixels 64..npix-1
#pragma unroll
          for (int i0 = 4 * CU; i0 < CTOT; i0 += 8) chain(i0, CTOT - i0 < 8 ? CTOT - i0 : 8);
          if (!MSUM && (npix & 1)) acc += prod[(lane < 5 ? lane : 4) * npad + npix - 1];
          __syncthreads();  // products consumed; next iteration may overwrite
          stamp(3);
        } else {
        // ---- I1 on the grid and I0 on its centre: one sample per lane and round (T:183-198, rows first).  All tap reads,
        // then all pixel reads, then the arithmetic, then the stores: LDS stores between the rounds would order every
        // later LDS read behind them (one LDS round trip per dependent access instead of three in total).
        {
          double2 fx[L::SR], fy[L::SR];
          int po[L::SR];
#pragma unroll
          for (int k = 0; k < L::SR; k++) {
            fx[k] = tapf[samp_cd[k]];
            fy[k] = tapf[32 + samp_rd[k]];
            po[k] = tapo[samp_cd[k]] + tapo[32 + samp_rd[k]];
          }
          float pv[L::SR][4];
#pragma unroll
          for (int k = 0; k < L::SR; k++) {
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(samp_i0[k] ? win0 : win1) + po[k]);
            pv[k][0] = p[0]; pv[k][1] = p[1]; pv[k][2] = p[KLT_PS]; pv[k][3] = p[KLT_PS + 1];
          }
          double val[L::SR];
#pragma unroll
          for (int k = 0; k < L::SR; k++) {
            const double v0 = (double)pv[k][0] * fx[k].y + (double)pv[k][1] * fx[k].x;
            const double v1 = (double)pv[k][2] * fx[k].y + (double)pv[k][3] * fx[k].x;
            val[k] = v0 * fy[k].y + v1 * fy[k].x;
          }
#pragma unroll
          for (int k = 0; k < L::SR; k++)
            if (samp_ok[k]) samp_out[k][0] = val[k];  // (only the last round has lanes without a sample)
        }
        __syncthreads();
        stamp(1);  // coordinate check + grid
        // ---- per-pixel products (T:440-449) from the grid
        const bool any_mis = (mis_p | mis_m) != 0ull;  // wave-uniform
        if (any_mis) ++slow_steps;
        {
          double gxp[PP], gxm[PP], gyp[PP], gym[PP], gcc[PP], irf[PP];
#pragma unroll
          for (int q = 0; q < PP; q++) {  // all grid reads first (see above)
            const int pix = lane + 64 * q < npix ? lane + 64 * q : 0;
            const int i = pix / side, j = pix % side;  // dy = i - r (outer), dx = j - r (inner)
            const double* gc1 = G1 + (i + 1) * GS + (j + 1);
            gxp[q] = gc1[1]; gxm[q] = gc1[-1]; gyp[q] = gc1[GS]; gym[q] = gc1[-GS]; gcc[q] = gc1[0];
            irf[q] = G0[(i + 1) * GS + (j + 1)];
          }
          if (any_mis) {  // neighbours that are not grid points: sampled with the taps the tap phase wrote for them
            auto lerp = [&](const double2& fx, const double2& fy, int off) {
              const float* p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(win1) + off);
              const double v0 = (double)p[0] * fx.y + (double)p[1] * fx.x;
              const double v1 = (double)p[KLT_PS] * fx.y + (double)p[KLT_PS + 1] * fx.x;
              return v0 * fy.y + v1 * fy.x;
            };
#pragma unroll
            for (int q = 0; q < PP; q++) {
              const int pix = lane + 64 * q < npix ? lane + 64 * q : 0;
              const int i = pix / side, j = pix % side;
              if ((mis_p >> j) & 1ull) gxp[q] = lerp(xtapf[j], tapf[32 + i + 1], xtapo[j] + tapo[32 + i + 1]);
              if ((mis_m >> j) & 1ull) gxm[q] = lerp(xtapf[32 + j], tapf[32 + i + 1], xtapo[32 + j] + tapo[32 + i + 1]);
              if ((mis_p >> (32 + i)) & 1ull) gyp[q] = lerp(tapf[j + 1], xtapf[16 + i], tapo[j + 1] + xtapo[16 + i]);
              if ((mis_m >> (32 + i)) & 1ull) gym[q] = lerp(tapf[j + 1], xtapf[48 + i], tapo[j + 1] + xtapo[48 + i]);
            }
          }
#pragma unroll
          for (int q = 0; q < PP; q++) {
            const int pix = lane + 64 * q;
            if (pix < npix) {
              const double Ix = 0.5 * (gxp[q] - gxm[q]), Iy = 0.5 * (gyp[q] - gym[q]), err = irf[q] - gcc[q];
              prod[0 * npad + pix] = Ix * Ix;
              prod[1 * npad + pix] = Ix * Iy;
              prod[2 * npad + pix] = Iy * Iy;
              prod[3 * npad + pix] = Ix * err;
              prod[4 * npad + pix] = Iy * err;
            }
          }
        }
        __syncthreads();
        stamp(2);  // products
        // ---- ordered sums
        if constexpr (MSUM) {
          // On the FP64 matrix core: v_mfma_f64_4x4x4 computes D[b][i][j] = C[b][i][j] + sum_k A[b][i][k] B[b][k][j] as four fused
          // multiply-adds in ascending k, each rounded to FP64 (measured on gfx950 over 1.28 M random sums: always the sequential
          // result, denormals kept, signed zeros as IEEE; tools/probes/mfma_f64_order.hip, profiles/r03_mfma_f64_order_probe.txt).
          // With B = 1.0 a product is exact and the instruction performs FOUR of the reference's ordered additions for 16
          // accumulators: acc_a <- (((acc_a + p[a][4s]) + p[a][4s+1]) + p[a][4s+2]) + p[a][4s+3].  Operand lane 16k + 4b + i holds
          // A[b][i][k]: accumulator a = 4b + i, addend 4s + k; rows a >= 5 read the zero row, the pad addends are +0.0 (an
          // accumulator that started at +0.0 is never -0.0, so x + (+0.0) = x).  D[b][i][j] sits in lane 16i + 4b + j.
          const int mk = lane >> 4, ma = lane & 15;
          const double* q = prod + (ma < 5 ? ma : 5) * npad + mk;
          // all operands are requested before the first instruction of the chain (the scheduler otherwise keeps two reads in
          // flight and the chain waits an LDS round trip for every other instruction)
          constexpr int NQ = npad / 4, CH = 32;
#pragma unroll
          for (int i0 = 0; i0 < NQ; i0 += CH) {
            double v[CH];
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < NQ) v[k] = q[4 * (i0 + k)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < NQ) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(v[k], 1.0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else
        if (lane < 5) {
          const double2* q = reinterpret_cast<const double2*>(prod + lane * npad);
          // compile-time trip count: the LDS reads are hoisted in batches ahead of the dependent adds
          constexpr int CH = 16;
#pragma unroll
          for (int i0 = 0; i0 < npix / 2; i0 += CH) {
            double2 v[CH];
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < npix / 2) v[k] = q[i0 + k];
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < npix / 2) { acc += v[k].x; acc += v[k].y; }
          }
          if (npix & 1) acc += prod[lane * npad + npix - 1];
        }
        __syncthreads();  // products consumed; next iteration may overwrite
        stamp(3);  // ordered sums
        }
        // accumulator a: lane a of the VALU chains, lane 16 (a & 3) + 4 (a >> 2) of the matrix-core result
        constexpr int LA0 = 0, LA1 = MSUM ? 16 : 1, LA2 = MSUM ? 32 : 2, LA3 = MSUM ? 48 : 3, LA4 = 4;
        const double A00 = readlane_f64(acc, LA0), A01 = readlane_f64(acc, LA1), A11 = readlane_f64(acc, LA2);
        const double b0 = readlane_f64(acc, LA3), b1 = readlane_f64(acc, LA4);
        // ---- 2x2 solve (T:451-459): inv00 = A11/detA, inv01 = -A01/detA, inv11 = A00/detA in lanes 0, 1, 2 of ONE division
        double sx = 0.0, sy = 0.0;
        const double detA = A00 * A11 - A01 * A01;
        if (!(fabs(detA) < 1e-9)) {
          const double num = lane == 0 ? A11 : (lane == 1 ? -A01 : A00);
          const double quo = num / detA;
          const double inv00 = readlane_f64(quo, 0), inv01 = readlane_f64(quo, 1), inv11 = readlane_f64(quo, 2);
          sx = inv00 * b0 + inv01 * b1;
          sy = inv01 * b0 + inv11 * b1;
        }
        ++steps;
        dlx += sx;
        dly += sy;
        stamp(4);  // 2x2 solve
        if (hypot_below(sx, sy, 1e-3)) break;
      }
      px = (plx + dlx) * (double)(1 << l);
      py = (ply + dly) * (double)(1 << l);
    }
    if (dir == 0 && lane == 0) {
      xy_fwd[2 * track] = px;
      xy_fwd[2 * track + 1] = py;
    }
  }
  if (lane == 0) {
    if (xy_back) {
      xy_back[2 * track] = px;
      xy_back[2 * track + 1] = py;
    }
    const double fb = sfmx::hypot_glibc(px - p0x, py - p0y);
    keep[track] = (fb >= fb_thresh) ? 0 : 1;  // T:362: `if (fb >= thresh) continue;`
    if (step_counter) step_counter[track] = (unsigned long long)steps | ((unsigned long long)slow_steps << 32);  // summed by the host
    if (STAMP && stamps) {
      for (int k = 0; k < 6; k++) stamps[8 * track + k] = tph[k];
      stamps[8 * track + 6] = steps;
      stamps[8 * track + 7] = slow_steps;
    }
  }
}

// ================================================================================================ K tracks per wavefront
// The step of one track keeps only 5 lanes busy in its ordered sums, 3 in its 2x2 solve and none in its loop control -- half of
// the ~4 000 cycles of an lk_step (profiles/r02_klt_stamps.txt).  k_klt_track_multi gives each of K tracks a GROUP of G = 64 / K
// lanes of one wavefront: the 121 add instructions of the ordered sums, the division sequence of the solve, the stop test and
// the loop control are issued once for all K tracks, and the sample / product phases run ceil(work / G) rounds.  Every
// arithmetic expression is the single-track kernel's (the reference's), only the lane that evaluates it changes.
//   * per-track state (estimate, level offset, window origin, activity) lives REPLICATED in the lanes of the track's group, so
//     the code reads like the one-track kernel; wave-uniform control flow (__any over the groups) decides what is executed;
//   * LDS per track: both staged windows as u8 (2 x 1.25 KB instead of 2 x 5.4 KB of f32: the conversion moves into the
//     sample, one unaligned 16-bit LDS read per pixel pair), the two sample grids, the products, the tap tables: 12 KB, i.e.
//     12 tracks per CU instead of 7;
//   * cross-lane traffic of the solve by DPP inside the group's first quad, the step back to the group through LDS.
template <int r>
struct KltLdsM {
  static constexpr int side = 2 * r + 1, npix = side * side, npad = (npix + 1) & ~1;
  static constexpr int NCAN = 2 * r + 3, GS = NCAN + 1;
  static constexpr int RS = 40;  // bytes per window row: 32 pixels + 8 (8-byte aligned rows, consecutive rows 10 banks apart)
  static constexpr size_t o_win0 = 0, o_win1 = (size_t)KLT_P * RS;
  static constexpr size_t o_g1 = (size_t)2 * KLT_P * RS;
  static constexpr size_t o_g0 = o_g1 + (size_t)GS * GS * 8;
  static constexpr size_t o_prod = o_g0 + (size_t)GS * GS * 8;
  static constexpr size_t o_tapf = (o_prod + (size_t)5 * npad * 8 + 15) & ~(size_t)15;  // double2 {f, 1-f} [2 axes][16]
  static constexpr size_t o_tapo = o_tapf + (size_t)2 * 16 * 16;                         // int [2][16]
  static constexpr size_t o_xtapf = o_tapo + (size_t)2 * 16 * 4;                         // double2 [2 kinds][2 axes][16]
  static constexpr size_t o_xtapo = o_xtapf + (size_t)2 * 2 * 16 * 16;                   // int [2][2][16]
  static constexpr size_t o_step = o_xtapo + (size_t)2 * 2 * 16 * 4;                     // double [2]: the step, for the whole group
  static constexpr size_t bytes = (o_step + 16 + 15) & ~(size_t)15;
  static constexpr int NS1 = NCAN * NCAN - 4;  // I1 on the grid without its four corners (no pixel has a diagonal neighbour)
  static constexpr int NS = NS1 + npix;        // + I0 on the grid's centre
};

typedef uint16_t __attribute__((aligned(1))) u16_unaligned;

template <int RS, int G>
__device__ __forceinline__ void stage_windows_u8(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int w, int h, int ox,
                                                 int oy, uint8_t* __restrict__ win0, uint8_t* __restrict__ win1, int gl) {
  if (ox >= 0 && oy >= 0 && ox + KLT_P <= w && oy + KLT_P <= h) {  // the usual case: 16-byte chunks, all loads before the stores
    constexpr int NC = 64 / G;
    uint32_t va[NC][4], vb[NC][4];
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int chunk = gl + G * c, row = chunk >> 1, c0 = (chunk & 1) * 16;
      const size_t off = (size_t)(oy + row) * w + ox + c0;
      const u32_unaligned* a = reinterpret_cast<const u32_unaligned*>(img0 + off);
      const u32_unaligned* b = reinterpret_cast<const u32_unaligned*>(img1 + off);
#pragma unroll
      for (int k = 0; k < 4; k++) { va[c][k] = a[k]; vb[c][k] = b[k]; }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
      const int chunk = gl + G * c, row = chunk >> 1, c0 = (chunk & 1) * 16;
      uint2* d0 = reinterpret_cast<uint2*>(win0 + row * RS + c0);
      uint2* d1 = reinterpret_cast<uint2*>(win1 + row * RS + c0);
      d0[0] = make_uint2(va[c][0], va[c][1]); d0[1] = make_uint2(va[c][2], va[c][3]);
      d1[0] = make_uint2(vb[c][0], vb[c][1]); d1[1] = make_uint2(vb[c][2], vb[c][3]);
    }
    return;
  }
  for (int idx = gl; idx < KLT_P * KLT_P; idx += G) {  // window over the image border: zero outside (never weighted, T:188)
    const int px = idx % KLT_P, py = idx / KLT_P;
    const int gx = ox + px, gy = oy + py;
    const bool ok = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const size_t off = ok ? (size_t)gy * w + gx : 0;
    const uint8_t a = img0[off], b = img1[off];
    win0[py * RS + px] = ok ? a : (uint8_t)0;
    win1[py * RS + px] = ok ? b : (uint8_t)0;
  }
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int r, int K, bool STAMP>
__global__ __launch_bounds__(64) void k_klt_track_multi(PyrDesc A, PyrDesc B, const double* __restrict__ xy_in, int n, int levels, int iters,
                                                        double fb_thresh, double* __restrict__ xy_fwd, double* __restrict__ xy_back,
                                                        uint8_t* __restrict__ keep, unsigned long long* __restrict__ step_counter,
                                                        unsigned long long* __restrict__ stamps, int wave_prio) {
  if (wave_prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (wave_prio == 2) __builtin_amdgcn_s_setprio(2);
  using L = KltLdsM<r>;
  constexpr int G = 64 / K;
  static_assert(K == 1 || K == 2 || K == 4, "a group is a whole number of 16-lane DPP rows");
  constexpr int side = L::side, npix = L::npix, npad = L::npad, NCAN = L::NCAN, GS = L::GS, RS = L::RS;
  static_assert(NCAN <= 16, "the grid slots of an axis sit in one DPP row");
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
  auto stamp = [&](int k) {
    if (STAMP) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tph[k] += now - tlast;
      tlast = now;
    }
  };
  if (STAMP) tlast = __builtin_amdgcn_s_memtime();
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x, g = lane / G, gl = lane % G;
  unsigned char* base = smem + (size_t)g * L::bytes;
  uint8_t* win0 = base + L::o_win0;
  uint8_t* win1 = base + L::o_win1;
  double* G1 = reinterpret_cast<double*>(base + L::o_g1);
  double* G0 = reinterpret_cast<double*>(base + L::o_g0);
  double* prod = reinterpret_cast<double*>(base + L::o_prod);
  double2* tapf = reinterpret_cast<double2*>(base + L::o_tapf);
  int* tapo = reinterpret_cast<int*>(base + L::o_tapo);
  double2* xtapf = reinterpret_cast<double2*>(base + L::o_xtapf);
  int* xtapo = reinterpret_cast<int*>(base + L::o_xtapo);
  double* stepv = reinterpret_cast<double*>(base + L::o_step);
  const int track = blockIdx.x * K + g;
  const bool valid = track < n;
  // the samples of this lane, fixed for the whole kernel: I1 at grid rows 1..side (all columns), then row 0 and row NCAN-1
  // (columns 1..side), then I0 at the centre points.  Packed: column slot | row slot << 5 | I0 << 10 | has-a-sample << 11.
  constexpr int SR = (L::NS + G - 1) / G;
  auto sample_desc = [&](int k) -> int {  // a handful of integer operations per sample; a table of them would cost SR registers
    int e = gl + G * k;
    const bool ok = e < L::NS;
    if (!ok) e = 0;
    int cd, rd, i0 = 0;
    if (e < side * NCAN) { rd = 1 + e / NCAN; cd = e % NCAN; }
    else if (e < side * NCAN + side) { rd = 0; cd = 1 + (e - side * NCAN); }
    else if (e < L::NS1) { rd = NCAN - 1; cd = 1 + (e - side * NCAN - side); }
    else { i0 = 1; rd = 1 + (e - L::NS1) / side; cd = 1 + (e - L::NS1) % side; }
    return cd | (rd << 5) | (i0 << 10) | ((ok ? 1 : 0) << 11);
  };
  const double p0x = valid ? xy_in[2 * track] : 0.0, p0y = valid ? xy_in[2 * track + 1] : 0.0;
  double px = p0x, py = p0y;
  unsigned int steps = 0, slow_steps = 0;
  // lane roles of the tap phase: a 16-lane row per axis.  G >= 32: row 0 of the group the x axis, row 1 the y axis, in one pass;
  // G == 16: the group's only row takes the axes one after the other
  constexpr int NPASS = G >= 32 ? 1 : 2;
  const int dt = gl & 15;
  const bool tap_role = G >= 32 ? gl < 32 : true;

  for (int dir = 0; dir < 2; ++dir) {
    for (int l = levels - 1; l >= 0; --l) {
      const uint8_t* img0 = dir == 0 ? A.px[l] : B.px[l];
      const uint8_t* img1 = dir == 0 ? B.px[l] : A.px[l];
      const int w = A.w[l], h = A.h[l];
      const double scale = 1.0 / (double)(1 << l);
      const double plx = px * scale, ply = py * scale;
      double dlx = 0.0, dly = 0.0;
      int ox = (int)0x7fffff00, oy = (int)0x7fffff00;  // no window staged yet
      bool active = valid;                             // this track still iterates on this level (uniform inside a group)
      for (int it = 0; it < iters; ++it) {
        if (!__any(active)) break;
        const double x = plx + dlx, y = ply + dly;
        stamp(5);  // level / loop bookkeeping
        const int bx = book_floor(x), by = book_floor(y);
        const bool touches = (bx + r + 3 >= 0) && (bx - r - 2 < w) && (by + r + 3 >= 0) && (by - r - 2 < h);
        if (active && !touches) {  // every sample is 0.0 (T:188) => step {0,0} (T:452) => the level ends (T:416); nothing to read
          ++steps;
          active = false;
        }
        {
          const bool covered = (bx - r - 2 >= ox) && (bx + r + 3 < ox + KLT_P) && (by - r - 2 >= oy) && (by + r + 3 < oy + KLT_P);
          const bool need = active && !covered;
          if (__any(need)) {
            if (need) {
              ox = bx - (KLT_P / 2 - 1);
              oy = by - (KLT_P / 2 - 1);
            }
            __syncthreads();
            if (need) stage_windows_u8<RS, G>(img0, img1, w, h, ox, oy, win0, win1, gl);
            __syncthreads();
          }
        }
        stamp(0);  // window staging
        // ---- coordinate taps of the grid slots and the bitwise on-grid test of the reference's neighbour coordinates (see
        // k_klt_track).  mis_*: bit d+r set = that neighbour of pixel offset d needs a sample of its own.
        unsigned mis_px = 0, mis_mx = 0, mis_py = 0, mis_my = 0;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          const int axis = G >= 32 ? ((gl >> 4) & 1) : pass;
          const double cv = slot_coord<r>(axis ? y : x, min(dt, NCAN - 1));
          const long long cbits = __double_as_longlong(cv);
          const long long next_bits = dpp_i64<0x101>(cbits);  // row_shl:1 = lane+1 (every lane enabled: see k_klt_track)
          const long long prev_bits = dpp_i64<0x111>(cbits);  // row_shr:1 = lane-1
          const bool inner = tap_role && dt >= 1 && dt <= side;
          const bool bad_p = inner & (__double_as_longlong(cv + 1) != next_bits);
          const bool bad_m = inner & (__double_as_longlong(cv - 1) != prev_bits);
          const unsigned long long bp = __ballot(bad_p), bm = __ballot(bad_m);
          constexpr unsigned M = (1u << side) - 1u;
          if (G >= 32) {
            mis_px = (unsigned)(bp >> (g * G + 1)) & M;  mis_mx = (unsigned)(bm >> (g * G + 1)) & M;
            mis_py = (unsigned)(bp >> (g * G + 17)) & M; mis_my = (unsigned)(bm >> (g * G + 17)) & M;
          } else if (pass == 0) {
            mis_px = (unsigned)(bp >> (g * G + 1)) & M;  mis_mx = (unsigned)(bm >> (g * G + 1)) & M;
          } else {
            mis_py = (unsigned)(bp >> (g * G + 1)) & M;  mis_my = (unsigned)(bm >> (g * G + 1)) & M;
          }
          if (tap_role && dt < NCAN) {
            const Tap tp = make_tap(cv, axis ? h : w, axis ? oy : ox);
            const bool ok = tp.m != 0.0;
            tapf[axis * 16 + dt] = make_double2(tp.f, ok ? 1 - tp.f : 0.0);
            tapo[axis * 16 + dt] = tp.l * (axis ? RS : 1);  // byte offset of the tap's row / column in a window
          }
          if (bad_p | bad_m) {
            const double ce = bad_p ? cv + 1 : cv - 1;
            const Tap tp = make_tap(ce, axis ? h : w, axis ? oy : ox);
            const int e = (bad_p ? 0 : 32) + axis * 16 + (dt - 1);
            xtapf[e] = make_double2(tp.f, tp.m != 0.0 ? 1 - tp.f : 0.0);
            xtapo[e] = tp.l * (axis ? RS : 1);
            if (bad_p & bad_m) {
              const Tap tq = make_tap(cv - 1, axis ? h : w, axis ? oy : ox);
              xtapf[32 + axis * 16 + (dt - 1)] = make_double2(tq.f, tq.m != 0.0 ? 1 - tq.f : 0.0);
              xtapo[32 + axis * 16 + (dt - 1)] = tq.l * (axis ? RS : 1);
            }
          }
        }
        __syncthreads();
        // ---- I1 on the grid and I0 on its centre (T:183-198, rows first), in batches of up to five samples per lane: all tap
        // reads, then all pixel reads, then the arithmetic, then the stores
        for (int k0 = 0; k0 < SR; k0 += 5) {
          constexpr int NBmax = 5;
          double2 fx[NBmax], fy[NBmax];
          int po[NBmax], sd[NBmax];
#pragma unroll
          for (int k = 0; k < NBmax; k++) {
            sd[k] = sample_desc(k0 + k < SR ? k0 + k : SR - 1);
            if (k0 + k >= SR) sd[k] &= ~(1 << 11);  // past the last round: computed, not stored
            const int cd = sd[k] & 31, rd = (sd[k] >> 5) & 31;
            fx[k] = tapf[cd];
            fy[k] = tapf[16 + rd];
            po[k] = tapo[cd] + tapo[16 + rd];
          }
          unsigned pa[NBmax], pb[NBmax];
#pragma unroll
          for (int k = 0; k < NBmax; k++) {
            const uint8_t* p = (((sd[k] >> 10) & 1) ? win0 : win1) + po[k];
            pa[k] = *reinterpret_cast<const u16_unaligned*>(p);
            pb[k] = *reinterpret_cast<const u16_unaligned*>(p + RS);
          }
          double val[NBmax];
#pragma unroll
          for (int k = 0; k < NBmax; k++) {
            const double v00 = (double)(pa[k] & 0xffu), v10 = (double)(pa[k] >> 8), v01 = (double)(pb[k] & 0xffu), v11 = (double)(pb[k] >> 8);
            const double v0 = v00 * fx[k].y + v10 * fx[k].x;
            const double v1 = v01 * fx[k].y + v11 * fx[k].x;
            val[k] = v0 * fy[k].y + v1 * fy[k].x;
          }
#pragma unroll
          for (int k = 0; k < NBmax; k++)
            if ((sd[k] >> 11) & 1) (((sd[k] >> 10) & 1) ? G0 : G1)[((sd[k] >> 5) & 31) * GS + (sd[k] & 31)] = val[k];
        }
        __syncthreads();
        stamp(1);  // coordinate check + grid
        // ---- per-pixel products (T:440-449) from the grid
        const bool any_mis = (mis_px | mis_mx | mis_py | mis_my) != 0u;  // uniform inside a group
        if (active && any_mis) ++slow_steps;
        constexpr int PP = (npix + G - 1) / G;
        for (int q0 = 0; q0 < PP; q0 += 4) {
          constexpr int QB = 4;
          double gxp[QB], gxm[QB], gyp[QB], gym[QB], gcc[QB], irf[QB];
#pragma unroll
          for (int q = 0; q < QB; q++)
            if (q0 + q < PP) {
              const int pix = gl + G * (q0 + q) < npix ? gl + G * (q0 + q) : 0;
              const int i = pix / side, j = pix % side;  // dy = i - r (outer), dx = j - r (inner)
              const double* gc1 = G1 + (i + 1) * GS + (j + 1);
              gxp[q] = gc1[1]; gxm[q] = gc1[-1]; gyp[q] = gc1[GS]; gym[q] = gc1[-GS]; gcc[q] = gc1[0];
              irf[q] = G0[(i + 1) * GS + (j + 1)];
            }
          if (any_mis) {  // neighbours that are not grid points: sampled with the taps the tap phase wrote for them
            auto lerp = [&](const double2& fxx, const double2& fyy, int off) {
              const uint8_t* p = win1 + off;
              const unsigned a = *reinterpret_cast<const u16_unaligned*>(p), b = *reinterpret_cast<const u16_unaligned*>(p + RS);
              const double v0 = (double)(a & 0xffu) * fxx.y + (double)(a >> 8) * fxx.x;
              const double v1 = (double)(b & 0xffu) * fxx.y + (double)(b >> 8) * fxx.x;
              return v0 * fyy.y + v1 * fyy.x;
            };
#pragma unroll
            for (int q = 0; q < QB; q++)
              if (q0 + q < PP) {
                const int pix = gl + G * (q0 + q) < npix ? gl + G * (q0 + q) : 0;
                const int i = pix / side, j = pix % side;
                if ((mis_px >> j) & 1u) gxp[q] = lerp(xtapf[j], tapf[16 + i + 1], xtapo[j] + tapo[16 + i + 1]);
                if ((mis_mx >> j) & 1u) gxm[q] = lerp(xtapf[32 + j], tapf[16 + i + 1], xtapo[32 + j] + tapo[16 + i + 1]);
                if ((mis_py >> i) & 1u) gyp[q] = lerp(tapf[j + 1], xtapf[16 + i], tapo[j + 1] + xtapo[16 + i]);
                if ((mis_my >> i) & 1u) gym[q] = lerp(tapf[j + 1], xtapf[48 + i], tapo[j + 1] + xtapo[48 + i]);
              }
          }
#pragma unroll
          for (int q = 0; q < QB; q++)
            if (q0 + q < PP) {
              const int pix = gl + G * (q0 + q);
              if (pix < npix) {
                const double Ix = 0.5 * (gxp[q] - gxm[q]), Iy = 0.5 * (gyp[q] - gym[q]), err = irf[q] - gcc[q];
                prod[0 * npad + pix] = Ix * Ix;
                prod[1 * npad + pix] = Ix * Iy;
                prod[2 * npad + pix] = Iy * Iy;
                prod[3 * npad + pix] = Ix * err;
                prod[4 * npad + pix] = Iy * err;
              }
            }
        }
        __syncthreads();
        stamp(2);  // products
        // ---- ordered sums: lane k < 5 of every group adds accumulator k's products in reference order -- one add instruction
        // serves the K tracks of the wave
        double acc = 0.0;
        if (gl < 5) {
          const double2* q = reinterpret_cast<const double2*>(prod + gl * npad);
          constexpr int CH = 16;
#pragma unroll
          for (int i0 = 0; i0 < npix / 2; i0 += CH) {
            double2 v[CH];
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < npix / 2) v[k] = q[i0 + k];
#pragma unroll
            for (int k = 0; k < CH; k++)
              if (i0 + k < npix / 2) { acc += v[k].x; acc += v[k].y; }
          }
          if (npix & 1) acc += prod[gl * npad + npix - 1];
        }
        stamp(3);  // ordered sums
        // ---- 2x2 solve (T:451-459) in the first quad of every group: the five sums by DPP, the three divisions in lanes 0..2
        // of ONE division sequence, the step back to the whole group through LDS
        const double A00 = dpp_f64<0x00>(acc), A01 = dpp_f64<0x55>(acc), A11 = dpp_f64<0xAA>(acc), b0 = dpp_f64<0xFF>(acc);
        const double b1 = dpp_f64<0x00>(dpp_f64<0x104>(acc));  // row_shl:4 brings lane 4's sum to lane 0, then to its quad
        const double detA = A00 * A11 - A01 * A01;
        const bool singular = fabs(detA) < 1e-9;
        const double num = gl == 0 ? A11 : (gl == 1 ? -A01 : A00);
        const double quo = num / detA;
        const double inv00 = dpp_f64<0x00>(quo), inv01 = dpp_f64<0x55>(quo), inv11 = dpp_f64<0xAA>(quo);
        const double sxq = singular ? 0.0 : inv00 * b0 + inv01 * b1;
        const double syq = singular ? 0.0 : inv01 * b0 + inv11 * b1;
        if (gl == 0) { stepv[0] = sxq; stepv[1] = syq; }
        __syncthreads();  // also: the products are consumed, the next step may overwrite them
        const double sx = stepv[0], sy = stepv[1];
        if (active) {
          ++steps;
          dlx += sx;
          dly += sy;
          if (hypot_below(sx, sy, 1e-3)) active = false;
        }
        stamp(4);  // 2x2 solve
      }
      px = (plx + dlx) * (double)(1 << l);
      py = (ply + dly) * (double)(1 << l);
    }
    if (dir == 0 && gl == 0 && valid) {
      xy_fwd[2 * track] = px;
      xy_fwd[2 * track + 1] = py;
    }
  }
  if (gl == 0 && valid) {
    if (xy_back) {
      xy_back[2 * track] = px;
      xy_back[2 * track + 1] = py;
    }
    const double fb = sfmx::hypot_glibc(px - p0x, py - p0y);
    keep[track] = (fb >= fb_thresh) ? 0 : 1;  // T:362: `if (fb >= thresh) continue;`
    if (step_counter) step_counter[track] = (unsigned long long)steps | ((unsigned long long)slow_steps << 32);  // summed by the host
    if (STAMP && stamps) {
      for (int k = 0; k < 6; k++) stamps[8 * track + k] = tph[k];
      stamps[8 * track + 6] = steps;
      stamps[8 * track + 7] = slow_steps;
    }
  }
}

template <int r>
static size_t klt_lds_bytes() { return KltLds<r>::bytes; }

extern "C" uint64_t sfmx_debug_klt_slow_steps(const sfmx_ctx* c) { return c ? c->klt_slow_steps : 0; }

extern "C" int sfmx_klt_track(sfmx_ctx* c, const sfmx_pyramid* pa, const sfmx_pyramid* pb, const double* xy_in, int n,
                              const sfmx_klt_cfg* cfg, double* xy_fwd, double* xy_back, uint8_t* keep, uint64_t* n_steps_out) {
  SFMX_REQUIRE(c, c && pa && pb && cfg && xy_fwd && keep && n >= 0);
  SFMX_REQUIRE(c, pa->w == pb->w && pa->h == pb->h && pa->levels == pb->levels);
  SFMX_REQUIRE(c, cfg->levels >= 1 && cfg->levels <= pa->levels && cfg->win_radius >= 1 && cfg->win_radius <= KLT_MAX_R && cfg->iters >= 0);
  if (n_steps_out) *n_steps_out = 0;
  if (int rc = sfmx_pyramid_settle(c, pa)) return rc;
  if (int rc = sfmx_pyramid_settle(c, pb)) return rc;
  if (n == 0) return SFMX_OK;
  SFMX_REQUIRE(c, xy_in != nullptr);
  const size_t nb = (size_t)n * 16;
  c->resident_points = 0;
  // No device slab and no DMA: the kernel reads the track positions straight out of pinned host memory (16 bytes per
  // wavefront) and writes fwd | back | per-track step counts | keep into a second pinned slab (one small posted write per
  // result and wavefront); the stream synchronisation below is the only wait.  The two copy kernels this replaces cost more
  // than the transfers themselves: ~10 us each plus their dispatch gaps, per call.
  const size_t o_fwd = 0, o_back = nb, o_steps = 2 * nb, o_keep = o_steps + (size_t)n * 8, out_bytes = o_keep + (size_t)n;
  SFMX_HIP(c, c->h[0].ensure(nb));
  SFMX_HIP(c, c->h[1].ensure(out_bytes));
  memcpy(c->h[0].p, xy_in, nb);
  char* dbase = c->h[1].as<char>();
  const int r = cfg->win_radius;
  KernelTimer t(c);
  t.start();
  prof_begin(c, KID_KLT);
  const int klt_prio = getenv("SFMX_KLT_WAVE_PRIO") ? atoi(getenv("SFMX_KLT_WAVE_PRIO")) : 0;  // read per call (A/B inside one process)
  static const bool stamps_on = getenv("SFMX_KLT_STAMPS") != nullptr;  // diagnostic build of the kernel, see k_klt_track
  unsigned long long* d_stamps = nullptr;
  if (stamps_on) {
    SFMX_HIP(c, c->d[1].ensure((size_t)n * 64));
    d_stamps = c->d[1].as<unsigned long long>();
    SFMX_HIP(c, hipMemsetAsync(d_stamps, 0, (size_t)n * 64, c->stream));
  }
#define KLT_ARGS                                                                                                                          \
  make_desc(pa), make_desc(pb), c->h[0].as<double>(), n, cfg->levels, cfg->iters, cfg->fb_thresh,                                         \
      reinterpret_cast<double*>(dbase + o_fwd), reinterpret_cast<double*>(dbase + o_back), reinterpret_cast<uint8_t*>(dbase + o_keep),  \
      reinterpret_cast<unsigned long long*>(dbase + o_steps), d_stamps, klt_prio
  // Tracks per wavefront.  One (K = 1) while the launch leaves SIMDs free anyway: a lone track per wave has the shortest
  // chain.  Two once there are more tracks than SIMDs (1 024 on MI355X): the waves of a SIMD then share its issue slots, and a
  // wave that carries two tracks issues the ordered sums, the solve and the loop control once for both.  SFMX_KLT_K = 0 (the
  // one-track kernel with f32 windows, the only one for win_radius 7), 1, 2, 4 overrides (A/B and tests; identical results).
  const int k_env = getenv("SFMX_KLT_K") ? atoi(getenv("SFMX_KLT_K")) : -1;  // read per call: the tests switch it inside one process
  static const int k2_min = getenv("SFMX_KLT_K2_MIN") ? atoi(getenv("SFMX_KLT_K2_MIN")) : 1024;
  int K = k_env >= 0 ? k_env : 0;  // measured (profiles/r03_klt_multi_probe.txt): the one-track kernel wins at every track count
  (void)k2_min;
  if (r > 6 || (K != 1 && K != 2 && K != 4)) K = 0;
#define KLT_LAUNCH_M(RR, KK)                                                                                                               \
  do {                                                                                                                                    \
    const size_t lds = (size_t)(KK) * KltLdsM<RR>::bytes;                                                                                 \
    if (stamps_on && RR == 5) k_klt_track_multi<RR, KK, true><<<(n + (KK)-1) / (KK), 64, lds, c->stream>>>(KLT_ARGS);                     \
    else k_klt_track_multi<RR, KK, false><<<(n + (KK)-1) / (KK), 64, lds, c->stream>>>(KLT_ARGS);                                         \
  } while (0)
#define KLT_LAUNCH_K(RR)                         \
  do {                                           \
    if (K == 1) KLT_LAUNCH_M(RR, 1);             \
    else if (K == 2) KLT_LAUNCH_M(RR, 2);        \
    else KLT_LAUNCH_M(RR, 4);                    \
  } while (0)
#define KLT_LAUNCH(RR)                                                                              \
  do {                                                                                              \
    if (stamps_on && RR == 5) k_klt_track<5, true><<<n, 64, klt_lds_bytes<5>(), c->stream>>>(KLT_ARGS); \
    else k_klt_track<RR, false><<<n, 64, klt_lds_bytes<RR>(), c->stream>>>(KLT_ARGS);               \
  } while (0)
#define KLT_LAUNCH_MSUM(RR)                                                                                              \
  do {                                                                                                                   \
    const size_t lds = KltLds<RR, true>::bytes + lds_pad;                                                                \
    if (stamps_on && RR == 5) k_klt_track<5, true, false, true><<<n, 64, KltLds<5, true>::bytes, c->stream>>>(KLT_ARGS);  \
    else k_klt_track<RR, false, false, true><<<n, 64, lds, c->stream>>>(KLT_ARGS);                                       \
  } while (0)
#define KLT_LAUNCH_PIPE_MSUM(RR)                                                                                             \
  do {                                                                                                                       \
    const size_t lds = KltLds<RR, true>::bytes + lds_pad;                                                                    \
    if (stamps_on && RR == 5) k_klt_track<5, true, true, true><<<n, 64, KltLds<5, true>::bytes, c->stream>>>(KLT_ARGS);       \
    else k_klt_track<RR, false, true, true><<<n, 64, lds, c->stream>>>(KLT_ARGS);                                            \
  } while (0)
#define KLT_LAUNCH_PIPE(RR)                                                                                        \
  do {                                                                                                             \
    if (stamps_on && RR == 5) k_klt_track<5, true, true><<<n, 64, klt_lds_bytes<5>(), c->stream>>>(KLT_ARGS);      \
    else k_klt_track<RR, false, true><<<n, 64, klt_lds_bytes<RR>(), c->stream>>>(KLT_ARGS);                        \
  } while (0)
  // Schedule of the ordered sums (identical results, measured in profiles/r03_klt_variants_probe.txt):
  //   SFMX_KLT_SUMS=mfma|valu  chains of FP64 matrix-core instructions (default from radius 2: 4-8 % shorter launches, and the
  //                            121 additions of a step leave the VALU) or of v_add_f64 in lanes 0..4;
  //   SFMX_KLT_PIPE=0|1        the sums over the first 64 pixels issued under the second half of the sample grid (radius 4 / 5);
  //                            default: with the matrix-core sums at radius 5 when the launch holds one to 2.5 waves per SIMD
  //                            (167 against 174 us at T = 1 564; slower for a lone wave and above ~3 waves per SIMD).
  // SFMX_KLT_LDS_PAD=bytes (experiment): extra dynamic LDS per workgroup of the matrix-core variants -- how much the other lanes'
  // kernels depend on the LDS the resident KLT waves leave free on a CU
  const size_t lds_pad = getenv("SFMX_KLT_LDS_PAD") ? (size_t)atoi(getenv("SFMX_KLT_LDS_PAD")) : 0;
  const char* sums_env = getenv("SFMX_KLT_SUMS");
  const char* pipe_env = getenv("SFMX_KLT_PIPE");
  const bool msum = sums_env ? sums_env[0] == 'm' : r >= 2;
  const bool pipe = pipe_env ? pipe_env[0] == '1' : (msum && r == 5 && n >= 1100 && n < 2600);
  if (K == 0 && msum && pipe && (r == 4 || r == 5)) {
    if (r == 4) KLT_LAUNCH_PIPE_MSUM(4);
    else KLT_LAUNCH_PIPE_MSUM(5);
  } else if (K == 0 && msum) {
    switch (r) {
      case 1: KLT_LAUNCH_MSUM(1); break;
      case 2: KLT_LAUNCH_MSUM(2); break;
      case 3: KLT_LAUNCH_MSUM(3); break;
      case 4: KLT_LAUNCH_MSUM(4); break;
      case 5: KLT_LAUNCH_MSUM(5); break;
      case 6: KLT_LAUNCH_MSUM(6); break;
      default: KLT_LAUNCH_MSUM(7); break;
    }
  } else if (K == 0) {
    switch (r) {
      case 1: KLT_LAUNCH(1); break;
      case 2: KLT_LAUNCH(2); break;
      case 3: KLT_LAUNCH(3); break;
      case 4: if (pipe) KLT_LAUNCH_PIPE(4); else KLT_LAUNCH(4); break;
      case 5: if (pipe) KLT_LAUNCH_PIPE(5); else KLT_LAUNCH(5); break;
      case 6: KLT_LAUNCH(6); break;
      default: KLT_LAUNCH(7); break;
    }
  } else {
    switch (r) {
      case 1: KLT_LAUNCH_K(1); break;
      case 2: KLT_LAUNCH_K(2); break;
      case 3: KLT_LAUNCH_K(3); break;
      case 4: KLT_LAUNCH_K(4); break;
      case 5: KLT_LAUNCH_K(5); break;
      default: KLT_LAUNCH_K(6); break;
    }
  }
#undef KLT_LAUNCH
#undef KLT_LAUNCH_PIPE
#undef KLT_LAUNCH_MSUM
#undef KLT_LAUNCH_PIPE_MSUM
#undef KLT_LAUNCH_K
#undef KLT_LAUNCH_M
#undef KLT_ARGS
  prof_end(c);
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  t.collect();
  const char* hp = dbase;  // [fwd nb][back nb][steps n*8][keep n]
  unsigned long long steps = 0, slow = 0;
  {
    const unsigned long long* ps = reinterpret_cast<const unsigned long long*>(hp + o_steps);
    for (int i = 0; i < n; i++) { steps += ps[i] & 0xffffffffull; slow += ps[i] >> 32; }
  }
  c->klt_slow_steps = slow;
  memcpy(xy_fwd, hp + o_fwd, nb);
  if (xy_back) memcpy(xy_back, hp + o_back, nb);
  memcpy(keep, hp + o_keep, (size_t)n);
  if (n_steps_out) *n_steps_out = steps;
  if (stamps_on) {  // per-phase s_memtime ticks per lk_step: track 0, the mean over all tracks, and the slowest track
    std::vector<unsigned long long> st((size_t)n * 8);
    SFMX_HIP(c, hipMemcpy(st.data(), d_stamps, (size_t)n * 64, hipMemcpyDeviceToHost));
    auto line = [&](const char* tag, const unsigned long long* v, double div) {
      const double k = v[6] ? 1.0 / (double)v[6] : 0.0;
      fprintf(stderr, "klt stamps %-8s (%6.0f steps, %5.0f off-grid): stage %5.0f | check+grid %5.0f | products %5.0f | sums %5.0f | solve %5.0f | loop %5.0f | total ticks %.0f\n",
              tag, v[6] / div, v[7] / div, v[0] * k, v[1] * k, v[2] * k, v[3] * k, v[4] * k, v[5] * k, (double)(v[0] + v[1] + v[2] + v[3] + v[4] + v[5]) / div);
    };
    unsigned long long sum[8] = {}, worst_total = 0;
    int worst = 0;
    for (int t = 0; t < n; t++) {
      unsigned long long tot = 0;
      for (int k = 0; k < 8; k++) { sum[k] += st[(size_t)8 * t + k]; if (k < 6) tot += st[(size_t)8 * t + k]; }
      if (tot > worst_total) { worst_total = tot; worst = t; }
    }
    line("track 0", &st[0], 1.0);
    line("mean", sum, (double)n);
    line("slowest", &st[(size_t)8 * worst], 1.0);
    fprintf(stderr, "klt stamps slowest track %d at (%.2f, %.2f)\n", worst, xy_in[2 * worst], xy_in[2 * worst + 1]);
  }
  return SFMX_OK;
}
