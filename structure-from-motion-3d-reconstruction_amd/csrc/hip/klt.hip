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
//    footprint of an lk_step leaves it.  Row stride 33 keeps the 11x11 access pattern off the
//    same bank.
//  * per-pixel phase (parallel): lane l owns window pixels l, l+64, ...; each pixel evaluates the six
//    bilinear samples of T:438-441 with the reference's exact expressions (floor, x - x0,
//    v00*(1-dx)+v10*dx, rows first) and the five products Ix*Ix, Ix*Iy, Iy*Iy, Ix*err, Iy*err.
//  * ordered reduction (serial by contract): FP64 addition is not associative and parity is
//    bit-exact, so the (2r+1)^2 products of each accumulator are added in the reference's
//    (dy outer, dx inner) sequence: lanes 0..4 each walk one accumulator's products in LDS.
//  * 2x2 solve, hypot-based stop test (glibc-compatible hypot, sfmx_math.h), level loop, then the
//    backward pass from the forward result and keep = !(hypot(back - p0) >= fb_thresh).
//
// No FMA contraction anywhere (-ffp-contract=off); FP64 division and sqrt are the correctly
// rounded forms.
#include "sfmx_internal.h"

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

// Both windows are staged together: all 32 byte loads of a lane are issued before the first LDS
// store, so one staging costs about one L2 round trip instead of 32 dependent ones.
template <int KLT_PS>
__device__ __forceinline__ void stage_windows(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int w, int h, int ox,
                                              int oy, float* __restrict__ win0, float* __restrict__ win1, int lane) {
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

template <int r>
__global__ __launch_bounds__(64) void k_klt_track(PyrDesc A, PyrDesc B, const double* __restrict__ xy_in, int n, int levels,
                                                  int iters, double fb_thresh, double* __restrict__ xy_fwd,
                                                  double* __restrict__ xy_back, uint8_t* __restrict__ keep,
                                                  unsigned long long* __restrict__ step_counter) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* win0 = reinterpret_cast<float*>(smem);                 // template image window (I0 of lk_step)
  float* win1 = win0 + KLT_P * KLT_PS_FOR(r);                   // current image window  (I1 of lk_step)
  double* prod = reinterpret_cast<double*>(win0 + ((2 * KLT_P * KLT_PS_FOR(r) + 3) & ~3));  // [5][npix_pad], 16-B aligned
  const int lane = threadIdx.x;
  const int track = blockIdx.x;
  if (track >= n) return;
  constexpr int side = 2 * r + 1, npix = side * side;
  constexpr int npad = (npix + 1) & ~1;
  constexpr int KLT_PS = KLT_PS_FOR(r);

  const double p0x = xy_in[2 * track], p0y = xy_in[2 * track + 1];
  double px = p0x, py = p0y;
  unsigned int steps = 0;

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
        // ---- make sure the staged window covers the footprint [b-r-2, b+r+3] of this step
        const int bx = book_floor(x), by = book_floor(y);
        const bool touches = (bx + r + 3 >= 0) && (bx - r - 2 < w) && (by + r + 3 >= 0) && (by - r - 2 < h);
        if (!touches) {
          // Every sample of this step is 0.0 in the reference (T:188) => A = 0, detA = 0 => step {0,0} (T:452)
          // => hypot(0,0) < 1e-3 ends the level (T:416).  Nothing to read: the window may be unstaged, and
          // stale LDS bits must never reach the mask-multiply in sample_lds (NaN * 0 = NaN).
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
        // ---- per-pixel products (T:433-449)
#pragma unroll
        for (int pix0 = 0; pix0 < npix; pix0 += 64) {
          const int pix = pix0 + lane;
          if (pix >= npix) break;
          const int dyi = pix / side - r, dxi = pix % side - r;
          const double xx = x + (double)dxi, yy = y + (double)dyi;
          const Tap cx0 = make_tap(xx, w, ox), cxp = make_tap(xx + 1, w, ox), cxm = make_tap(xx - 1, w, ox);
          const Tap cy0 = make_tap(yy, h, oy), cyp = make_tap(yy + 1, h, oy), cym = make_tap(yy - 1, h, oy);
          const double Ix = 0.5 * (sample_lds<KLT_PS>(win1, cxp, cy0) - sample_lds<KLT_PS>(win1, cxm, cy0));
          const double Iy = 0.5 * (sample_lds<KLT_PS>(win1, cx0, cyp) - sample_lds<KLT_PS>(win1, cx0, cym));
          const double Iref = sample_lds<KLT_PS>(win0, cx0, cy0);
          const double Icur = sample_lds<KLT_PS>(win1, cx0, cy0);
          const double err = Iref - Icur;
          prod[0 * npad + pix] = Ix * Ix;
          prod[1 * npad + pix] = Ix * Iy;
          prod[2 * npad + pix] = Iy * Iy;
          prod[3 * npad + pix] = Ix * err;
          prod[4 * npad + pix] = Iy * err;
        }
        __syncthreads();
        // ---- ordered sums: lane k < 5 adds accumulator k's products in reference order
        double acc = 0.0;
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
        const double A00 = readlane_f64(acc, 0), A01 = readlane_f64(acc, 1), A11 = readlane_f64(acc, 2);
        const double b0 = readlane_f64(acc, 3), b1 = readlane_f64(acc, 4);
        // ---- 2x2 solve (T:451-459)
        double sx = 0.0, sy = 0.0;
        const double detA = A00 * A11 - A01 * A01;
        if (!(fabs(detA) < 1e-9)) {
          const double inv00 = A11 / detA, inv01 = -A01 / detA, inv11 = A00 / detA;
          sx = inv00 * b0 + inv01 * b1;
          sy = inv01 * b0 + inv11 * b1;
        }
        ++steps;
        dlx += sx;
        dly += sy;
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
    if (step_counter) atomicAdd(step_counter, (unsigned long long)steps);
  }
}

extern "C" int sfmx_klt_track(sfmx_ctx* c, const sfmx_pyramid* pa, const sfmx_pyramid* pb, const double* xy_in, int n,
                              const sfmx_klt_cfg* cfg, double* xy_fwd, double* xy_back, uint8_t* keep, uint64_t* n_steps_out) {
  SFMX_REQUIRE(c, c && pa && pb && cfg && xy_fwd && keep && n >= 0);
  SFMX_REQUIRE(c, pa->w == pb->w && pa->h == pb->h && pa->levels == pb->levels);
  SFMX_REQUIRE(c, cfg->levels >= 1 && cfg->levels <= pa->levels && cfg->win_radius >= 1 && cfg->win_radius <= KLT_MAX_R && cfg->iters >= 0);
  if (n_steps_out) *n_steps_out = 0;
  if (n == 0) return SFMX_OK;
  SFMX_REQUIRE(c, xy_in != nullptr);
  const size_t nb = (size_t)n * 16;
  c->resident_points = 0;
  // one device slab: [xy_in nb][fwd nb][back nb][steps 8][keep n]; one pinned slab for the upload and one
  // for the download, so a call costs two DMA transfers and one host synchronisation
  // device slab [xy_in nb][steps 8 (+8 pad)][fwd nb][back nb][keep n]: the upload covers the inputs and the zeroed
  // step counter, the download everything behind it
  const size_t o_steps = nb, o_fwd = nb + 16, o_back = o_fwd + nb, o_keep = o_back + nb, dev_bytes = o_keep + (size_t)n;
  SFMX_HIP(c, c->d[0].ensure(dev_bytes + 64));
  SFMX_HIP(c, c->h[0].ensure(nb + 16));
  SFMX_HIP(c, c->h[1].ensure(dev_bytes - nb));
  char* dbase = c->d[0].as<char>();
  memcpy(c->h[0].p, xy_in, nb);
  memset(c->h[0].as<char>() + nb, 0, 16);
  SFMX_HIP(c, hipMemcpyAsync(dbase, c->h[0].p, nb + 16, hipMemcpyHostToDevice, c->stream));
  const int r = cfg->win_radius, npix = (2 * r + 1) * (2 * r + 1), npad = (npix + 1) & ~1;
  const size_t shmem = (size_t)((2 * KLT_P * KLT_PS_FOR(r) + 3) & ~3) * sizeof(float) + (size_t)5 * npad * sizeof(double);
  KernelTimer t(c);
  t.start();
#define KLT_LAUNCH(RR)                                                                                                          \
  k_klt_track<RR><<<n, 64, shmem, c->stream>>>(make_desc(pa), make_desc(pb), reinterpret_cast<double*>(dbase), n, cfg->levels, cfg->iters, \
                                               cfg->fb_thresh, reinterpret_cast<double*>(dbase + o_fwd), reinterpret_cast<double*>(dbase + o_back), \
                                               reinterpret_cast<uint8_t*>(dbase + o_keep), reinterpret_cast<unsigned long long*>(dbase + o_steps))
  switch (r) {
    case 1: KLT_LAUNCH(1); break;
    case 2: KLT_LAUNCH(2); break;
    case 3: KLT_LAUNCH(3); break;
    case 4: KLT_LAUNCH(4); break;
    case 5: KLT_LAUNCH(5); break;
    case 6: KLT_LAUNCH(6); break;
    default: KLT_LAUNCH(7); break;
  }
#undef KLT_LAUNCH
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, dbase + o_steps, dev_bytes - nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  t.collect();
  const char* hp = c->h[1].as<char>();  // [steps 16][fwd nb][back nb][keep n]
  unsigned long long steps = 0;
  memcpy(&steps, hp, 8);
  memcpy(xy_fwd, hp + 16, nb);
  if (xy_back) memcpy(xy_back, hp + 16 + nb, nb);
  memcpy(keep, hp + 16 + 2 * nb, (size_t)n);
  if (n_steps_out) *n_steps_out = steps;
  return SFMX_OK;
}
