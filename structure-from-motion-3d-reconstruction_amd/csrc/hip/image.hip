// image.hip — pyramid build (T:200-232) and Shi-Tomasi score map (T:242-272) for gfx950.
//
// Both are HBM-streaming kernels: every pixel is read once from HBM (neighbours come from L2/L1)
// and each output is written once.  Algorithmic bytes:
//   downsample level l:  read w_l*h_l, write w_l*h_l/4
//   score map:           read w*h (u8), write 8*w*h (f64)
#include "sfmx_internal.h"

#include <cstdlib>
#include <list>
#include <mutex>
#include <utility>
#include <vector>

// ------------------------------------------------------------------------------------------ pyramid
// One thread per output pixel; 2x2 box, integer sum / 4 (truncation), +1 neighbours clamped.
__global__ void k_downsample2(const uint8_t* __restrict__ src, int sw, int sh, uint8_t* __restrict__ dst, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const int sx = 2 * x, sy = 2 * y;
  const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
  const int sum = (int)src[(size_t)sy * sw + sx] + (int)src[(size_t)sy * sw + sx1] + (int)src[(size_t)sy1 * sw + sx] +
                  (int)src[(size_t)sy1 * sw + sx1];
  dst[(size_t)y * dw + x] = (uint8_t)(sum / 4);
}

static int build_levels(sfmx_ctx* c, sfmx_pyramid* p, hipStream_t stream) {
  const bool prof = stream == c->stream;
  if (prof && p->levels > 1) prof_begin(c, KID_PYRAMID);  // all levels of one image = one profile entry
  struct End { sfmx_ctx* c; bool on; ~End() { if (on) prof_end(c); } } end{c, prof};
  for (int l = 1; l < p->levels; l++) {
    if (p->lw[l] <= 0 || p->lh[l] <= 0) continue;
    dim3 b(64, 4), g((p->lw[l] + 63) / 64, (p->lh[l] + 3) / 4);
    k_downsample2<<<g, b, 0, stream>>>(p->base + p->off[l - 1], p->lw[l - 1], p->lh[l - 1], p->base + p->off[l], p->lw[l], p->lh[l]);
    SFMX_HIP(c, hipGetLastError());
  }
  return SFMX_OK;
}
// a pyramid that is still being built on the second stream: the context's main stream waits for it (device side, no host wait)
int sfmx_pyramid_settle(sfmx_ctx* c, const sfmx_pyramid* p) {
  sfmx_pyramid* q = const_cast<sfmx_pyramid*>(p);
  if (q->ready_pending) {
    SFMX_HIP(c, hipStreamWaitEvent(c->stream, q->ready, 0));
    q->ready_pending = false;
  }
  return SFMX_OK;
}

extern "C" {

int sfmx_pyramid_create(sfmx_ctx* c, int w, int h, int levels, sfmx_pyramid** out) {
  SFMX_REQUIRE(c, c && out && w > 0 && h > 0 && levels >= 1 && levels <= SFMX_MAX_LEVELS);
  sfmx_pyramid* p = new sfmx_pyramid;
  p->w = w;
  p->h = h;
  p->levels = levels;
  size_t off = 0;
  int lw = w, lh = h;
  for (int l = 0; l < levels; l++) {
    p->lw[l] = lw;
    p->lh[l] = lh;
    p->off[l] = off;
    off += (((size_t)lw * lh) + 255) & ~(size_t)255;
    lw /= 2;
    lh /= 2;
  }
  p->bytes = off + 256;
  hipError_t e = hipMalloc((void**)&p->base, p->bytes);
  if (e != hipSuccess) {
    delete p;
    return sfmx_fail(c, SFMX_ERR_HIP, "hipMalloc(pyramid)", e);
  }
  *out = p;
  return SFMX_OK;
}

void sfmx_pyramid_destroy(sfmx_ctx* c, sfmx_pyramid* p) {
  if (!p) return;
  if (c) { (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->copy_stream); }
  if (p->ready) (void)hipEventDestroy(p->ready);
  p->fetched.release();
  if (p->base) (void)hipFree(p->base);
  delete p;
}

int sfmx_pyramid_upload(sfmx_ctx* c, sfmx_pyramid* p, const uint8_t* host_pixels) {
  SFMX_REQUIRE(c, c && p && host_pixels);
  if (int rc = sfmx_pyramid_settle(c, p)) return rc;
  p->fetched_level = -1;
  SFMX_HIP(c, hipMemcpyAsync(p->base, host_pixels, (size_t)p->w * p->h, hipMemcpyHostToDevice, c->stream));
  return build_levels(c, p, c->stream);
}

int sfmx_pyramid_set_device(sfmx_ctx* c, sfmx_pyramid* p, const void* device_pixels) {
  SFMX_REQUIRE(c, c && p && device_pixels);
  if (int rc = sfmx_pyramid_settle(c, p)) return rc;
  p->fetched_level = -1;
  SFMX_HIP(c, hipMemcpyAsync(p->base, device_pixels, (size_t)p->w * p->h, hipMemcpyDeviceToDevice, c->stream));
  return build_levels(c, p, c->stream);
}

// The same on the context's SECOND stream, so that the pyramid of the next frame is built while the kernels of the current
// one (KLT) occupy the first; optionally the pixels of one level (the 32x32 descriptor's source) travel to pinned host memory
// in the same go.  The caller must not have work in flight that still reads this pyramid object.  sfmx_pyramid_wait orders the
// context's main stream behind the build; sfmx_pyramid_fetched_level waits on the host for the copy and returns the pixels.
int sfmx_pyramid_set_device_async(sfmx_ctx* c, sfmx_pyramid* p, const void* device_pixels, int fetch_level) {
  SFMX_REQUIRE(c, c && p && device_pixels && fetch_level < p->levels);
  if (!p->ready) SFMX_HIP(c, hipEventCreateWithFlags(&p->ready, hipEventDisableTiming));
  SFMX_HIP(c, hipMemcpyAsync(p->base, device_pixels, (size_t)p->w * p->h, hipMemcpyDeviceToDevice, c->copy_stream));
  const int rc = build_levels(c, p, c->copy_stream);
  if (rc) return rc;
  p->fetched_level = -1;
  if (fetch_level >= 0) {
    const size_t nb = (size_t)p->lw[fetch_level] * p->lh[fetch_level];
    SFMX_HIP(c, p->fetched.ensure(nb > 0 ? nb : 1));
    if (nb) SFMX_HIP(c, hipMemcpyAsync(p->fetched.p, p->base + p->off[fetch_level], nb, hipMemcpyDeviceToHost, c->copy_stream));
    p->fetched_level = fetch_level;
  }
  SFMX_HIP(c, hipEventRecord(p->ready, c->copy_stream));
  p->ready_pending = true;
  return SFMX_OK;
}
int sfmx_pyramid_wait(sfmx_ctx* c, sfmx_pyramid* p) {
  SFMX_REQUIRE(c, c && p);
  return sfmx_pyramid_settle(c, p);
}
int sfmx_pyramid_fetched_level(sfmx_ctx* c, sfmx_pyramid* p, int level, const uint8_t** out) {
  SFMX_REQUIRE(c, c && p && out);
  *out = nullptr;
  if (p->fetched_level != level || !p->ready) return SFMX_OK;  // not fetched: the caller downloads
  SFMX_HIP(c, hipEventSynchronize(p->ready));
  *out = p->fetched.as<uint8_t>();
  return SFMX_OK;
}

int sfmx_pyramid_download_level(sfmx_ctx* c, const sfmx_pyramid* p, int level, uint8_t* host_out) {
  SFMX_REQUIRE(c, c && p && host_out && level >= 0 && level < p->levels);
  if (int rc = sfmx_pyramid_settle(c, p)) return rc;
  const size_t nb = (size_t)p->lw[level] * p->lh[level];
  if (nb == 0) return SFMX_OK;
  SFMX_HIP(c, hipMemcpyAsync(host_out, p->base + p->off[level], nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  return SFMX_OK;
}

int sfmx_pyramid_level_size(const sfmx_pyramid* p, int level, int* w, int* h) {
  if (!p || level < 0 || level >= p->levels) return SFMX_ERR_INVALID;
  if (w) *w = p->lw[level];
  if (h) *h = p->lh[level];
  return SFMX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ Shi-Tomasi
// Tile = 64 x 8 output pixels per 256-thread... (512-thread) block.  The u8 tile plus a 3-pixel halo
// (2 for the 5x5 box + 1 for the central difference, indices clamped exactly like T:242-249) is
// staged in LDS once; gradients are exact halves of small integers, so gx*gx etc. and the 25-term
// sums are exact in FP64 and the only rounding steps are tr*tr-4*det and the (correctly rounded)
// sqrt — evaluated in the reference's order.
#define ST_TX 64
#define ST_TY 8
#define ST_HALO 3
#define ST_LW (ST_TX + 2 * ST_HALO)
#define ST_LH (ST_TY + 2 * ST_HALO)

__global__ __launch_bounds__(ST_TX* ST_TY) void k_shi_score(const uint8_t* __restrict__ img, int w, int h,
                                                            double* __restrict__ score, unsigned long long* __restrict__ max_bits) {
  __shared__ uint8_t tile[ST_LH][ST_LW + 2];
  __shared__ unsigned long long wave_max[(ST_TX * ST_TY) / 64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int x0 = blockIdx.x * ST_TX, y0 = blockIdx.y * ST_TY;
  const int tid = ty * ST_TX + tx;
  // LDS holds the image value at CLAMPED coordinates, so tile[yy][xx] == im.at(clamp) for every
  // index the gradient lambdas can form.
  for (int i = tid; i < ST_LH * ST_LW; i += ST_TX * ST_TY) {
    const int ly = i / ST_LW, lx = i % ST_LW;
    const int gx = min(max(x0 + lx - ST_HALO, 0), w - 1);
    const int gy = min(max(y0 + ly - ST_HALO, 0), h - 1);
    tile[ly][lx] = img[(size_t)gy * w + gx];
  }
  __syncthreads();
  const int x = x0 + tx, y = y0 + ty;
  double s = 0.0;
  const bool inside = (x < w && y < h);
  if (inside && x >= 2 && x < w - 2 && y >= 2 && y < h - 2) {
    double sxx = 0, sxy = 0, syy = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const int ly = ty + ST_HALO + dy, lx = tx + ST_HALO + dx;
        // clamping of xm/xp/ym/yp (T:243,247) is already folded into the tile contents, EXCEPT that
        // the lambdas clamp the neighbour index, not the centre: centre (xx,yy) is always in-image here.
        const double gx = 0.5 * ((double)tile[ly][lx + 1] - (double)tile[ly][lx - 1]);
        const double gy = 0.5 * ((double)tile[ly + 1][lx] - (double)tile[ly - 1][lx]);
        sxx += gx * gx;
        sxy += gx * gy;
        syy += gy * gy;
      }
    }
    const double tr = sxx + syy;
    const double det = sxx * syy - sxy * sxy;
    const double d0 = tr * tr - 4.0 * det;
    const double disc = (0.0 < d0) ? d0 : 0.0;  // std::max(0.0, d0)
    s = 0.5 * (tr - sqrt(disc));
  }
  if (inside) score[(size_t)y * w + x] = s;
  // block max of the (non-negative) scores via their bit patterns
  unsigned long long b = inside ? (unsigned long long)__double_as_longlong(s) : 0ull;
  if (s < 0.0 || s != s) b = 0ull;  // cannot happen for exact inputs; keeps the reduction monotone
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_down(b, o, 64);
    b = t > b ? t : b;
  }
  if ((tid & 63) == 0) wave_max[tid >> 6] = b;
  __syncthreads();
  if (tid == 0) {
    unsigned long long m = wave_max[0];
    for (int i = 1; i < (ST_TX * ST_TY) / 64; i++) m = wave_max[i] > m ? wave_max[i] : m;
    atomicMax(max_bits, m);
  }
}

// Ordered compaction of candidates (score >= thr) in row-major order: one block per image row
// counts, a single-block scan turns counts into offsets, a second pass writes.  Row-major order is
// part of the contract (it is the std::sort input order at T:286).
__global__ void k_row_count(const double* __restrict__ score, int w, int h, const unsigned long long* __restrict__ max_bits,
                            double quality, int* __restrict__ row_count) {
  const int y = blockIdx.x;
  const double thr = __longlong_as_double((long long)*max_bits) * quality;
  int c = 0;
  for (int x = threadIdx.x; x < w; x += blockDim.x) c += (score[(size_t)y * w + x] >= thr) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  __shared__ int part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += part[i];
    row_count[y] = t;
  }
}
__global__ void k_row_scan(int* __restrict__ row_count, int h, int* __restrict__ total) {
  // exclusive scan by one wave-strided thread block (h <= a few thousand rows)
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < h; base += 64) {
    const int i = base + (int)threadIdx.x;
    int v = (i < h) ? row_count[i] : 0;
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if ((int)threadIdx.x >= o) incl += t;
    }
    const int c = carry;
    if (i < h) row_count[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 63) carry = c + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void k_row_write(const double* __restrict__ score, int w, int h, const unsigned long long* __restrict__ max_bits,
                            double quality, const int* __restrict__ row_off, int cap, uint32_t* __restrict__ cand_xy,
                            double* __restrict__ cand_s) {
  // one wave per row keeps the in-row order with a ballot prefix
  const int y = blockIdx.x;
  const double thr = __longlong_as_double((long long)*max_bits) * quality;
  int off = row_off[y];
  for (int base = 0; base < w; base += 64) {
    const int x = base + (int)threadIdx.x;
    const double s = (x < w) ? score[(size_t)y * w + x] : -1.0;
    const bool hit = (x < w) && (s >= thr);
    const unsigned long long m = __ballot(hit);
    if (hit) {
      const int pos = off + __popcll(m & ((1ull << threadIdx.x) - 1ull));
      if (pos < cap) {
        cand_xy[pos] = (uint32_t)x | ((uint32_t)y << 16);
        cand_s[pos] = s;
      }
    }
    off += __popcll(m);
  }
}

// ---- certain-outcome resolution of the greedy min-distance pick (T:288-300) --------------------------
// The greedy pick visits candidates by descending score and accepts one iff no ACCEPTED corner lies
// closer than min_dist.  Whatever order the sort gives equal scores, two facts hold:
//   (A) a candidate is certainly ACCEPTED once every other pixel within min_dist whose score is >= its
//       own is certainly rejected (nothing that could precede it can block it);
//   (R) a candidate is certainly REJECTED once a certainly-accepted pixel of strictly greater score lies
//       within min_dist; a rejected candidate never influences a later decision.
// Both are monotone, so they can be applied in parallel and in place, round after round (a parallel
// fixpoint of the sequential greedy).  After a few rounds almost every candidate is decided; only the
// accepted ones and the few still-undecided ones (equal-score neighbours, long dependency chains) travel
// to the host, which finishes with the reference's own sort + pick on that short list (pipeline.cpp).
// state: 0 = below threshold, 1 = undecided candidate, 2 = accepted, 3 = rejected.
__global__ __launch_bounds__(256) void k_shi_init(const double* __restrict__ score, int w, int h,
                                                  const unsigned long long* __restrict__ max_bits, double quality,
                                                  uint8_t* __restrict__ state, int* __restrict__ zero_ints, int n_zero) {
  // the work-list counters of the later sweeps start at zero (this used to be a fill launch of its own)
  if (zero_ints && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < n_zero) zero_ints[threadIdx.x] = 0;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const double thr = __longlong_as_double((long long)*max_bits) * quality;
  state[(size_t)y * w + x] = (score[(size_t)y * w + x] >= thr) ? 1 : 0;
}
// One round over a 64x4-pixel tile per 256-thread block (one pixel per thread, ~1200 blocks for VGA so
// that ~19 waves per CU hide the LDS latency of the tap loop).  Blocks without an undecided pixel leave
// after one byte load per pixel; the others stage scores + states of the tile and its (min_dist-1)
// halo in LDS once, so the neighbourhood scans are LDS traffic (the global-memory version of this scan
// was a ~50 us dependent-load chain per undecided pixel).
#define SR_TX 64
#define SR_TY 4
#define SR_MAXR 15
#define SR_MAXACC 96  // accepted pixels are >= min_dist apart: a staged (64+2r) x (4+2r) area holds far fewer
// One sweep of rules (R) and (A) over a 64x4 tile, evaluated from a snapshot of the tile and its halo in LDS:
//   (R) a candidate is rejected if an ACCEPTED pixel of its disc has a strictly greater score -- the accepted pixels of
//       the staged area are collected into a short list while staging (a few dozen at most), so this is a loop over
//       that list instead of a scan of the 15x15 neighbourhood;
//   (A) it is accepted if no live (undecided or accepted) pixel of its disc has a score >= its own.  Almost every
//       candidate is ruled out by one of its eight direct neighbours (the score field is smooth), so each thread tests
//       those first; only the 3x3 maxima that are left (a few per tile) get the full disc, sixteen lanes per pixel.
// Same decisions as scanning the whole disc per pixel (the first version: 450 LDS reads per undecided pixel, 56 us per
// sweep over a VGA image), at a tenth of the LDS traffic.
__global__ __launch_bounds__(256) void k_shi_round(const double* __restrict__ score, int w, int h, int md, uint8_t* __restrict__ state,
                                                   int* __restrict__ changed, int inner) {
  extern __shared__ __align__(16) unsigned char sr_mem[];
  __shared__ int n_acc, n_sur;
  __shared__ unsigned short acc_idx[SR_MAXACC], sur_idx[256];
  __shared__ unsigned char sur_blocked[256];
  const int r = md - 1, md2 = md * md;
  const int lw = SR_TX + 2 * r, lh = SR_TY + 2 * r;
  double* ts = reinterpret_cast<double*>(sr_mem);                  // [lh][lw] scores
  uint8_t* tq = reinterpret_cast<uint8_t*>(ts + (size_t)lw * lh);  // [lh][lw] states
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int x0 = blockIdx.x * SR_TX, y0 = blockIdx.y * SR_TY;
  const int x = x0 + tx, y = y0 + ty;
  const uint8_t mine = (x < w && y < h) ? state[(size_t)y * w + x] : 0;
  if (tid == 0) { n_acc = 0; n_sur = 0; }
  if (!__syncthreads_or(mine == 1)) return;
  for (int i = tid; i < lw * lh; i += 256) {
    const int ly = i / lw, lx = i - ly * lw;
    const int gx = x0 + lx - r, gy = y0 + ly - r;
    const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const uint8_t st = in ? state[(size_t)gy * w + gx] : 0;
    ts[i] = in ? score[(size_t)gy * w + gx] : -1.0;
    tq[i] = st;
    if (st == 2) {
      const int k = atomicAdd(&n_acc, 1);
      if (k < SR_MAXACC) acc_idx[k] = (unsigned short)i;
    }
  }
  __syncthreads();
  const int cy = ty + r, cx = tx + r;
  const double s = ts[cy * lw + cx];
  uint8_t st_mine = mine;
  bool wrote = false;
  // `inner` sweeps on the staged tile: a decision taken in one sweep is visible (LDS state tile, accepted list) to the
  // block's other pixels in the next, so dependency chains that stay inside the tile resolve within one launch and one
  // staging.  Halo states stay at their snapshot, which is always safe.
  for (int sweep = 0; sweep < inner; ++sweep) {
    const int nacc = n_acc;
    const bool listed = nacc <= SR_MAXACC && r >= 1;  // otherwise: the plain scan of the whole disc
    bool rejected = false, blocked = false;
    if (tid == 0) n_sur = 0;
    __syncthreads();
    if (st_mine == 1) {
      if (listed) {
        for (int k = 0; k < nacc; k++) {  // (R)
          const int i = acc_idx[k];
          const int ay = i / lw, ax = i - ay * lw;
          const int dx = ax - cx, dy = ay - cy;
          rejected |= (dx * dx + dy * dy < md2) && (ts[i] > s);
        }
        if (!rejected) {  // (A), direct neighbours first (all inside the disc: md >= 2)
#pragma unroll
          for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
              if (dx == 0 && dy == 0) continue;
              const uint8_t st = tq[(cy + dy) * lw + cx + dx];
              blocked |= (st == 1 || st == 2) && (ts[(cy + dy) * lw + cx + dx] >= s);
            }
          if (!blocked) {
            const int k = atomicAdd(&n_sur, 1);
            sur_idx[k] = (unsigned short)tid;
            sur_blocked[tid] = 0;
          }
        }
      } else {
        for (int dy = -r; dy <= r && !rejected; ++dy) {
          const int rem = md2 - dy * dy;  // dx*dx < rem
          const double* rs = ts + (cy + dy) * lw + cx;
          const uint8_t* rq = tq + (cy + dy) * lw + cx;
          for (int dx = -r; dx <= r; ++dx) {
            const uint8_t st = rq[dx];
            const double sq = rs[dx];
            const bool rel = (dx * dx < rem) && !(dx == 0 && dy == 0) && (st == 1 || st == 2);
            rejected |= rel && (st == 2) && (sq > s);  // (R)
            blocked |= rel && (sq >= s);
          }
        }
      }
    }
    __syncthreads();
    if (listed) {  // the 3x3 maxima: the whole disc, sixteen lanes per pixel
      const int nsur = n_sur, side = 2 * r + 1, ntap = side * side;
      const int g = tid >> 4, l16 = tid & 15, gsh = (g & 3) * 16;
      for (int base = 0; base < nsur; base += 16) {
        const int e = base + g;
        const bool live = e < nsur;
        const int t = live ? sur_idx[e] : 0;
        const int ccx = (t & 63) + r, ccy = (t >> 6) + r;
        const double ss = ts[ccy * lw + ccx];
        bool hit = false;
        if (live)
          for (int k = l16; k < ntap; k += 16) {
            const int dy = k / side - r, dx = k - (k / side) * side - r;
            const uint8_t st = tq[(ccy + dy) * lw + ccx + dx];
            const double sq = ts[(ccy + dy) * lw + ccx + dx];
            hit |= (dx * dx + dy * dy < md2) && !(dx == 0 && dy == 0) && (st == 1 || st == 2) && (sq >= ss);
          }
        const unsigned any = (unsigned)((__ballot(hit) >> gsh) & 0xffffull);
        if (live && l16 == 0 && any) sur_blocked[t] = 1;
      }
      __syncthreads();
      if (st_mine == 1 && !rejected && !blocked) blocked = sur_blocked[tid] != 0;
    }
    // every scan of this sweep has read the tile: decisions become visible
    const bool decided = st_mine == 1 && (rejected || !blocked);
    const int any_decided = __syncthreads_or(decided);
    if (decided) {
      st_mine = rejected ? 3 : 2;
      tq[cy * lw + cx] = st_mine;
      wrote = true;
      if (st_mine == 2) {
        const int k = atomicAdd(&n_acc, 1);
        if (k < SR_MAXACC) acc_idx[k] = (unsigned short)(cy * lw + cx);
      }
    }
    if (!any_decided) break;
    __syncthreads();
  }
  if (wrote) {
    state[(size_t)y * w + x] = st_mine;
    *changed = 1;
  }
}

// ---- tile-resident fixpoint ----------------------------------------------------------------------------
// The dense sweeps above pay a staging of scores + states per sweep and the list sweeps a full disc scan per undecided pixel
// per sweep (33 MB of traffic and ~20 launches per VGA image).  k_shi_tile keeps a TW x TH tile and its (min_dist - 1) halo in
// LDS for a whole PASS and iterates rules (A) / (R) there until nothing inside the tile changes:
//   * every undecided pixel keeps a WITNESS -- one live pixel of its disc whose score is >= its own, the strongest found.  While
//     the witness is live the pixel cannot be accepted, so a round costs it two LDS reads; an accepted witness of strictly
//     greater score rejects it (R).  Only a pixel whose witness has been rejected looks for a new one: its eight neighbours
//     first (the score field is smooth), the whole disc only if they have none (sixteen lanes per pixel); no witness at all
//     = every pixel of the disc with a score >= its own is rejected = accepted (A);
//   * a newly accepted pixel STAMPS its disc: every undecided pixel of strictly smaller score is rejected at once (R from the
//     accepted side), which is what makes the dependency chains short.
// Halo pixels keep the state they had when the tile was staged (a decision is a certain fact, an old state only says less:
// any mix of old and new states is sound) and are owned by the neighbouring tile; what stays undecided along tile borders is
// taken up by the next pass, whose tile grid is shifted by half a tile.  After the last pass the undecided pixels travel to the
// host with the accepted ones, exactly as before.  state: 0 below threshold, 1 undecided, 2 accepted, 3 rejected.
#define SHT_TW 64
#define SHT_TH 32
#define SHT_THREADS 512
#define SHT_NONE 0xffffu
__global__ __launch_bounds__(SHT_THREADS) void k_shi_tile(const double* __restrict__ score, int w, int h, int md, uint8_t* __restrict__ state,
                                                          const unsigned long long* __restrict__ max_bits, double quality, int first_pass,
                                                          int offx, int offy, const int8_t* __restrict__ taps_g, int ntaps, int max_rounds) {
  extern __shared__ __align__(16) unsigned char sht_mem[];
  __shared__ int n_sur, n_acc, n_acc_done;
  const int r = md - 1, md2 = md * md;
  const int lw = SHT_TW + 2 * r, lh = SHT_TH + 2 * r, la = lw * lh;
  double* ts = reinterpret_cast<double*>(sht_mem);                                    // [lh][lw] scores (-1 outside the image)
  unsigned short* wit = reinterpret_cast<unsigned short*>(ts + la);                   // [TH][TW] witness: index into the staged area
  unsigned short* sur = wit + SHT_TW * SHT_TH;                                        // [TW*TH] pixels that need the whole disc
  unsigned short* acc = sur + SHT_TW * SHT_TH;                                        // [la] accepted pixels whose disc is to be stamped
  uint8_t* tq = reinterpret_cast<uint8_t*>(acc + la);                                 // [lh][lw] states
  int8_t* taps = reinterpret_cast<int8_t*>(tq + ((la + 15) & ~15));                   // [ntaps][2] offsets of the open disc
  const int tid = threadIdx.x;
  const int x0 = (int)blockIdx.x * SHT_TW - offx, y0 = (int)blockIdx.y * SHT_TH - offy;
  constexpr int PPT = SHT_TW * SHT_TH / SHT_THREADS;  // interior pixels per thread
  // ---- is there anything to decide in this tile?  (first pass: states do not exist yet)
  if (!first_pass) {
    bool und = false;
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const int i = tid + k * SHT_THREADS, gx = x0 + (i % SHT_TW), gy = y0 + (i / SHT_TW);
      und |= gx >= 0 && gx < w && gy >= 0 && gy < h && state[(size_t)gy * w + gx] == 1;
    }
    if (!__syncthreads_or(und)) return;
  }
  if (tid == 0) { n_sur = 0; n_acc = 0; n_acc_done = 0; }
  for (int i = tid; i < 2 * ntaps; i += SHT_THREADS) taps[i] = taps_g[i];
  const double thr = __longlong_as_double((long long)*max_bits) * quality;
  __syncthreads();
  // ---- stage scores and states of the tile and its halo; accepted pixels go on the stamp list
  for (int i = tid; i < la; i += SHT_THREADS) {
    const int ly = i / lw, lx = i - ly * lw;
    const int gx = x0 + lx - r, gy = y0 + ly - r;
    const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const double sc = in ? score[(size_t)gy * w + gx] : -1.0;
    uint8_t st;
    if (first_pass) st = (in && sc >= thr) ? 1 : 0;
    else st = in ? state[(size_t)gy * w + gx] : 0;
    ts[i] = sc;
    tq[i] = st;
    if (st == 2) acc[atomicAdd(&n_acc, 1)] = (unsigned short)i;
  }
  for (int i = tid; i < SHT_TW * SHT_TH; i += SHT_THREADS) wit[i] = SHT_NONE;
  __syncthreads();
  const int g16 = tid >> 4, l16 = tid & 15, gsh = (g16 & 3) * 16;
  for (int round = 0; round < max_rounds; ++round) {
    // ---- (R) stamp the discs of the pixels accepted since the last stamp
    const int a0 = n_acc_done, a1 = n_acc;
    __syncthreads();
    if (tid == 0) { n_acc_done = a1; n_sur = 0; }
    for (int e = a0 + g16; e < a1; e += SHT_THREADS / 16) {
      const int ai = acc[e], ay = ai / lw, ax = ai - ay * lw;
      const double sa = ts[ai];
      for (int k = l16; k < ntaps; k += 16) {
        const int qx = ax + taps[2 * k], qy = ay + taps[2 * k + 1];
        if (qx < 0 || qx >= lw || qy < 0 || qy >= lh) continue;
        const int qi = qy * lw + qx;
        if (tq[qi] == 1 && ts[qi] < sa) tq[qi] = 3;
      }
    }
    __syncthreads();
    // ---- witnesses: keep, reject by, or replace (eight neighbours first)
    bool changed = a1 > a0;
#pragma unroll
    for (int k = 0; k < PPT; k++) {
      const int i = tid + k * SHT_THREADS;
      const int cy = i / SHT_TW + r, cx = i % SHT_TW + r, ci = cy * lw + cx;
      if (tq[ci] != 1) continue;
      const double s = ts[ci];
      const unsigned wi = wit[i];
      if (wi != SHT_NONE) {
        const uint8_t ws = tq[wi];
        if (ws == 1) continue;  // blocked by a live pixel: nothing to do this round
        if (ws == 2) {
          if (ts[wi] > s) { tq[ci] = 3; changed = true; }  // (R)
          continue;  // an accepted pixel of EQUAL score blocks for good (a tie: the host decides)
        }
      }
      // the witness is gone (or there never was one): strongest live neighbour with a score >= s
      int best = -1;
      double bs = -1.0;
      if (r >= 1) {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            if (dx * dx + dy * dy >= md2) continue;  // min_dist 2: the diagonal neighbours are outside the disc
            const int qi = ci + dy * lw + dx;
            const uint8_t st = tq[qi];
            const double sq = ts[qi];
            if ((st == 1 || st == 2) && sq >= s && sq > bs) { bs = sq; best = qi; }
          }
      }
      if (best >= 0) {
        wit[i] = (unsigned short)best;
        changed = true;
        if (tq[best] == 2 && bs > s) tq[ci] = 3;
      } else {
        sur[atomicAdd(&n_sur, 1)] = (unsigned short)i;
      }
    }
    __syncthreads();
    // ---- the whole disc for the pixels without a neighbouring witness, sixteen lanes per pixel
    const int nsur = n_sur;
    for (int base = 0; base < nsur; base += SHT_THREADS / 16) {
      const int e = base + g16;
      const bool live = e < nsur;
      const int i = live ? sur[e] : 0;
      const int cy = i / SHT_TW + r, cx = i % SHT_TW + r, ci = cy * lw + cx;
      const double s = ts[ci];
      int best = -1;
      double bs = -1.0;
      if (live)
        for (int k = l16; k < ntaps; k += 16) {
          const int qi = ci + taps[2 * k + 1] * lw + taps[2 * k];  // inside the staged area: the halo is as wide as the disc
          const uint8_t st = tq[qi];
          const double sq = ts[qi];
          if ((st == 1 || st == 2) && sq >= s && sq > bs) { bs = sq; best = qi; }
        }
      // strongest blocker over the sixteen lanes (score, then index, so that every lane agrees)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const double os = __shfl_xor(bs, o, 16);
        const int ob = __shfl_xor(best, o, 16);
        if (os > bs || (os == bs && ob > best)) { bs = os; best = ob; }
      }
      if (live && l16 == 0) {
        if (best >= 0) {
          wit[i] = (unsigned short)best;
          if (tq[best] == 2 && bs > s) tq[ci] = 3;
        } else {
          tq[ci] = 2;  // (A): nothing live in the disc reaches its score
          acc[atomicAdd(&n_acc, 1)] = (unsigned short)ci;
        }
      }
      (void)gsh;
    }
    changed |= nsur > 0;
    if (!__syncthreads_or(changed)) break;
  }
  __syncthreads();
  // ---- the tile's own pixels go back (the halo belongs to the neighbours)
#pragma unroll
  for (int k = 0; k < PPT; k++) {
    const int i = tid + k * SHT_THREADS;
    const int gx = x0 + (i % SHT_TW), gy = y0 + (i / SHT_TW);
    if (gx >= 0 && gx < w && gy >= 0 && gy < h) state[(size_t)gy * w + gx] = tq[(i / SHT_TW + r) * lw + (i % SHT_TW) + r];
  }
}
static size_t shi_tile_lds(int md, int ntaps) {
  const int r = md - 1;
  const size_t la = (size_t)(SHT_TW + 2 * r) * (SHT_TH + 2 * r);
  return la * 8 + (size_t)SHT_TW * SHT_TH * 2 * 2 + la * 2 + ((la + 15) & ~(size_t)15) + (size_t)2 * ntaps + 32;
}

// ---- work-list sweeps -------------------------------------------------------------------------------
// After the first (tiled, all-pixel) sweep about a third of the candidates are still undecided and the
// dependency chains need ~30 more sweeps.  Those run on a compact list of undecided pixel indices:
// 16 lanes cooperate on one pixel (taps of its min_dist disc come from a small offset table, ~13 per
// lane, all loads in flight together), decide by a 16-lane ballot, and re-append the pixel to the next
// list if it is still undecided.  Cost per sweep is proportional to the number of undecided pixels.
// One pixel's (R)/(A) evidence gathered by 16 cooperating lanes: 8 taps per lane are in flight at a time
// (state and score are loaded unconditionally from a safe address, so nothing serialises on a branch).
__device__ __forceinline__ void shi_scan_taps(const double* __restrict__ score, const uint8_t* __restrict__ state, int w, int h,
                                              const int8_t* __restrict__ taps, int ntaps, int lane16, int x, int y, uint32_t idx, double s,
                                              bool& rejected, bool& blocked) {
  for (int t0 = lane16; t0 < ntaps; t0 += 16 * 8) {
    uint32_t off[8];
    bool in[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int t = t0 + 16 * k;
      const bool valid = t < ntaps;
      const int xx = x + (valid ? (int)taps[2 * t] : 0), yy = y + (valid ? (int)taps[2 * t + 1] : 0);
      in[k] = valid && xx >= 0 && xx < w && yy >= 0 && yy < h;
      off[k] = in[k] ? (uint32_t)yy * (uint32_t)w + (uint32_t)xx : idx;
    }
    uint8_t st[8];
    double sq[8];
#pragma unroll
    for (int k = 0; k < 8; k++) st[k] = state[off[k]];
#pragma unroll
    for (int k = 0; k < 8; k++) sq[k] = score[off[k]];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const bool rel = in[k] && (st[k] == 1 || st[k] == 2);
      rejected |= rel && (st[k] == 2) && (sq[k] > s);  // (R)
      blocked |= rel && (sq[k] >= s);
    }
  }
}

__global__ __launch_bounds__(256) void k_shi_list_build(const uint8_t* __restrict__ state, int n, uint32_t* __restrict__ list, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool und = (i < n) && state[i] == 1;
  const unsigned long long m = __ballot(und);
  int base = 0;
  if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(count, (int)__popcll(m));
  base = __shfl(base, 0, 64);
  if (und) list[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void k_shi_list_sweep(const double* __restrict__ score, int w, int h, uint8_t* __restrict__ state,
                                                        const int8_t* __restrict__ taps, int ntaps, const uint32_t* __restrict__ list_in,
                                                        const int* __restrict__ count_in, uint32_t* __restrict__ list_out,
                                                        int* __restrict__ count_out) {
  const int n = *count_in;
  const int lane16 = threadIdx.x & 15, group = threadIdx.x >> 4;        // 16 groups of 16 lanes per block
  const int gsh = ((threadIdx.x >> 4) & 3) * 16;                         // position of the group inside its wave
  for (int e0 = blockIdx.x * 16; e0 < n; e0 += gridDim.x * 16) {
    const int e = e0 + group;
    const bool live = e < n;
    const uint32_t idx = live ? list_in[e] : 0u;
    const int y = (int)(idx / (uint32_t)w), x = (int)(idx - (uint32_t)y * (uint32_t)w);
    const double s = live ? score[idx] : 0.0;
    bool rejected = false, blocked = false;
    if (live) shi_scan_taps(score, state, w, h, taps, ntaps, lane16, x, y, idx, s, rejected, blocked);
    const unsigned rj = (unsigned)((__ballot(rejected) >> gsh) & 0xffffull);
    const unsigned bl = (unsigned)((__ballot(blocked) >> gsh) & 0xffffull);
    const bool still = live && !rj && bl;
    if (live && lane16 == 0) {
      if (rj) state[idx] = 3;
      else if (!bl) state[idx] = 2;                      // (A)
    }
    // re-append the undecided ones (one atomic per wave)
    const unsigned long long m = __ballot(still && lane16 == 0);
    int base = 0;
    if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(count_out, (int)__popcll(m));
    base = __shfl(base, 0, 64);
    if (still && lane16 == 0) list_out[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = idx;
  }
}

// Tail: once the list is short, ONE 1024-thread workgroup runs the remaining sweeps back to back (global
// writes of a sweep are visible to the whole workgroup after the barrier), instead of one ~6 us launch
// per sweep.  Works for any list length (the block strides over it); meant for a few hundred entries.
__global__ __launch_bounds__(1024) void k_shi_list_tail(const double* __restrict__ score, int w, int h, uint8_t* __restrict__ state,
                                                        const int8_t* __restrict__ taps, int ntaps, uint32_t* __restrict__ list_a,
                                                        uint32_t* __restrict__ list_b, const int* __restrict__ count_in, int max_sweeps) {
  __shared__ int s_count[2];
  const int lane16 = threadIdx.x & 15, group = threadIdx.x >> 4;  // 64 groups
  const int gsh = ((threadIdx.x >> 4) & 3) * 16;
  if (threadIdx.x == 0) { s_count[0] = *count_in; s_count[1] = 0; }
  __syncthreads();
  uint32_t* lin = list_a;
  uint32_t* lout = list_b;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    const int cur = sweep & 1;
    const int n = s_count[cur];
    if (n == 0) break;
    for (int e0 = 0; e0 < n; e0 += 64) {
      const int e = e0 + group;
      const bool live = e < n;
      const uint32_t idx = live ? lin[e] : 0u;
      const int y = (int)(idx / (uint32_t)w), x = (int)(idx - (uint32_t)y * (uint32_t)w);
      const double s = live ? score[idx] : 0.0;
      bool rejected = false, blocked = false;
      if (live) shi_scan_taps(score, state, w, h, taps, ntaps, lane16, x, y, idx, s, rejected, blocked);
      const unsigned rj = (unsigned)((__ballot(rejected) >> gsh) & 0xffffull);
      const unsigned bl = (unsigned)((__ballot(blocked) >> gsh) & 0xffffull);
      const bool still = live && !rj && bl;
      if (live && lane16 == 0) {
        if (rj) state[idx] = 3;
        else if (!bl) state[idx] = 2;
      }
      const unsigned long long m = __ballot(still && lane16 == 0);
      int base = 0;
      if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&s_count[cur ^ 1], (int)__popcll(m));
      base = __shfl(base, 0, 64);
      if (still && lane16 == 0) lout[base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_count[cur] = 0;  // becomes the output counter of the next sweep
    uint32_t* tmp = lin; lin = lout; lout = tmp;
    __syncthreads();
  }
}

// ordered compaction (row-major): survivors (state 1 or 2) with their score and their index in the list
// of ALL candidates (state != 0), plus that full list's scores (kept resident for the tie-order replay)
__global__ void k_flag_row_count(const uint8_t* __restrict__ flag, int w, int* __restrict__ row_count, int* __restrict__ row_all) {
  const int y = blockIdx.x;
  int c = 0, a = 0;
  for (int x = threadIdx.x; x < w; x += blockDim.x) {
    const uint8_t f = flag[(size_t)y * w + x];
    c += (f == 1 || f == 2) ? 1 : 0;
    a += (f != 0) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_down(c, o, 64); a += __shfl_down(a, o, 64); }
  __shared__ int part[4], parta[4];
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = c; parta[threadIdx.x >> 6] = a; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0, ta = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) { t += part[i]; ta += parta[i]; }
    row_count[y] = t;
    row_all[y] = ta;
  }
}
__global__ void k_flag_row_write(const double* __restrict__ score, const uint8_t* __restrict__ flag, int w, const int* __restrict__ row_off,
                                 const int* __restrict__ row_off_all, int cap, uint32_t* __restrict__ cand_xy, double* __restrict__ cand_s,
                                 int32_t* __restrict__ cand_full, double2* __restrict__ all_keys, const unsigned long long* __restrict__ header,
                                 char* __restrict__ pin, int spec) {
  // pin (optional, pinned HOST memory): [0,16) the header {max score bits, #survivors, #candidates}, then the first `spec`
  // survivors as [xy u32][score f64][full index i32] -- written from here instead of through four DMA copies
  const size_t o_xy = 64, o_s = o_xy + (size_t)spec * 4, o_full = o_s + (size_t)spec * 8;
  uint32_t* pin_xy = reinterpret_cast<uint32_t*>(pin + o_xy);
  double* pin_s = reinterpret_cast<double*>(pin + o_s);
  int32_t* pin_full = reinterpret_cast<int32_t*>(pin + o_full);
  const int y = blockIdx.x;
  if (pin && y == 0 && threadIdx.x < 2) reinterpret_cast<unsigned long long*>(pin)[threadIdx.x] = header[threadIdx.x];
  int off = row_off[y], offa = row_off_all[y];
  for (int base = 0; base < w; base += 64) {
    const int x = base + (int)threadIdx.x;
    const uint8_t f = (x < w) ? flag[(size_t)y * w + x] : 0;
    const bool hit = (f == 1 || f == 2), any = (f != 0);
    const unsigned long long m = __ballot(hit), ma = __ballot(any);
    const unsigned long long below = (1ull << threadIdx.x) - 1ull;
    const int posa = offa + __popcll(ma & below);
    const double s = any ? score[(size_t)y * w + x] : 0.0;
    if (any) all_keys[posa] = make_double2(s, __hiloint2double(0, posa));  // {score, (id = posa, mark = 0)}: the host's SortKey
    if (hit) {
      const int pos = off + __popcll(m & below);
      if (pos < cap) {
        const uint32_t packed = (uint32_t)x | ((uint32_t)y << 16) | (f == 2 ? 0x80000000u : 0u);
        cand_xy[pos] = packed;
        cand_s[pos] = s;
        cand_full[pos] = posa;
        if (pin && pos < spec) {
          pin_xy[pos] = packed;
          pin_s[pos] = s;
          pin_full[pos] = posa;
        }
      }
    }
    off += __popcll(m);
    offa += __popcll(ma);
  }
}

static int launch_score(sfmx_ctx* c, const sfmx_pyramid* p, double* d_score, unsigned long long* d_max) {
  if (int rc = sfmx_pyramid_settle(c, p)) return rc;
  SFMX_HIP(c, hipMemsetAsync(d_max, 0, 8, c->stream));
  dim3 b(ST_TX, ST_TY), g((p->w + ST_TX - 1) / ST_TX, (p->h + ST_TY - 1) / ST_TY);
  KernelTimer t(c);
  t.start();
  SFMX_PROF(c, KID_SHI_SCORE, (k_shi_score<<<g, b, 0, c->stream>>>(p->base + p->off[0], p->w, p->h, d_score, d_max)));
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  return SFMX_OK;
}

extern "C" {

int sfmx_shi_tomasi_score(sfmx_ctx* c, const sfmx_pyramid* p, double* score_out, double* max_out) {
  SFMX_REQUIRE(c, c && p && (score_out || max_out));
  const size_t n = (size_t)p->w * p->h;
  c->resident_points = 0;
  SFMX_HIP(c, c->d[0].ensure(n * 8));
  SFMX_HIP(c, c->d[1].ensure(64));
  int rc = launch_score(c, p, c->d[0].as<double>(), c->d[1].as<unsigned long long>());
  if (rc) return rc;
  double mx = 0;
  if (score_out) SFMX_HIP(c, hipMemcpyAsync(score_out, c->d[0].p, n * 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(&mx, c->d[1].p, 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  KernelTimer(c).collect();
  if (max_out) *max_out = mx;
  return SFMX_OK;
}

int sfmx_shi_tomasi_candidates(sfmx_ctx* c, const sfmx_pyramid* p, double quality, int cap, uint32_t* cand_xy,
                               double* cand_score, int* n_out, double* max_out) {
  SFMX_REQUIRE(c, c && p && cand_xy && cand_score && n_out && cap > 0 && p->w < 65536 && p->h < 65536);
  const size_t n = (size_t)p->w * p->h;
  c->resident_points = 0;
  SFMX_HIP(c, c->d[0].ensure(n * 8));
  SFMX_HIP(c, c->d[1].ensure(64));
  SFMX_HIP(c, c->d[2].ensure((size_t)(p->h + 1) * 4 + 64));
  SFMX_HIP(c, c->d[3].ensure((size_t)cap * 4));
  SFMX_HIP(c, c->d[4].ensure((size_t)cap * 8));
  unsigned long long* d_max = c->d[1].as<unsigned long long>();
  int* d_rows = c->d[2].as<int>();
  int* d_total = d_rows + p->h;
  int rc = launch_score(c, p, c->d[0].as<double>(), d_max);
  if (rc) return rc;
  k_row_count<<<p->h, 256, 0, c->stream>>>(c->d[0].as<double>(), p->w, p->h, d_max, quality, d_rows);
  k_row_scan<<<1, 64, 0, c->stream>>>(d_rows, p->h, d_total);
  k_row_write<<<p->h, 64, 0, c->stream>>>(c->d[0].as<double>(), p->w, p->h, d_max, quality, d_rows, cap, c->d[3].as<uint32_t>(),
                                          c->d[4].as<double>());
  SFMX_HIP(c, hipGetLastError());
  int total = 0;
  double mx = 0;
  SFMX_HIP(c, hipMemcpyAsync(&total, d_total, 4, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(&mx, d_max, 8, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  KernelTimer(c).collect();
  const int m = total < cap ? total : cap;
  if (m > 0) {  // large (MB-sized) download: through pinned staging, pageable D2H is several times slower
    SFMX_HIP(c, c->h[0].ensure((size_t)m * 4));
    SFMX_HIP(c, c->h[1].ensure((size_t)m * 8));
    SFMX_HIP(c, hipMemcpyAsync(c->h[0].p, c->d[3].p, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, c->d[4].p, (size_t)m * 8, hipMemcpyDeviceToHost, c->stream));
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(cand_xy, c->h[0].p, (size_t)m * 4);
    memcpy(cand_score, c->h[1].p, (size_t)m * 8);
  }
  *n_out = total;
  if (max_out) *max_out = mx;
  return SFMX_OK;
}

// The whole device side of the call -- score map, 18 fixpoint rounds, ordered compaction and the
// speculative download into pinned memory -- is ~30 launches of 2-30 us kernels: launch-bound when issued
// one by one (~10 us of host time each).  It is captured ONCE per (buffers, image, parameters) signature
// into a hipGraph and replayed afterwards: one graph launch + one synchronisation per call.
// schedule of the corner fixpoint for an image: number of tile passes (0 = the sweep schedule) and the tile kernel's inner rounds.
// SFMX_SHI_MODE=tile[,passes[,rounds]] | sweeps overrides the choice by image size; read per call (tests / A/B inside one process)
static int shi_tile_passes(int w, int h, int* rounds_out) {
  int mode_passes = -1, tile_rounds = 64;  // -1: by image size
  if (const char* e = getenv("SFMX_SHI_MODE")) {
    int a = 3, b = 64;
    if (strncmp(e, "tile", 4) == 0) {
      mode_passes = 3;
      if (sscanf(e, "tile,%d,%d", &a, &b) >= 1) { mode_passes = a < 1 ? 1 : (a > 16 ? 16 : a); tile_rounds = b < 1 ? 1 : b; }
    } else if (strncmp(e, "sweeps", 6) == 0) {
      mode_passes = 0;
    }
  }
  if (rounds_out) *rounds_out = tile_rounds;
  (void)w; (void)h;
  return mode_passes >= 0 ? mode_passes : 3;  // the tile-resident kernel, three passes, at every image size (see below)
}
struct ShiGraphKey {
  const void* img; void* d0; void* d1; void* d2; void* d3; void* d4; void* d5; void* d6; void* pin; void* w0; void* w1; void* w2; void* w3;
  int w, h, md, cap, tile_passes, tile_rounds;
  double quality;
  bool operator==(const ShiGraphKey& o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct ShiGraph { ShiGraphKey key; hipGraphExec_t exec; };
// per-context caches live in one registry; contexts belong to different host threads, hence the lock
static std::mutex g_graph_mu;
static std::list<std::pair<sfmx_ctx*, std::vector<ShiGraph>>> g_graph_registry;
static std::vector<ShiGraph>& shi_graphs(sfmx_ctx* c) {
  std::lock_guard<std::mutex> lk(g_graph_mu);
  for (auto& e : g_graph_registry) if (e.first == c) return e.second;
  g_graph_registry.emplace_back(c, std::vector<ShiGraph>());
  return g_graph_registry.back().second;  // std::list: the reference stays valid when other contexts register
}
void sfmx_release_graphs(sfmx_ctx* c) {
  std::lock_guard<std::mutex> lk(g_graph_mu);
  for (auto it = g_graph_registry.begin(); it != g_graph_registry.end(); ++it)
    if (it->first == c) {
      for (auto& g : it->second) (void)hipGraphExecDestroy(g.exec);
      g_graph_registry.erase(it);
      return;
    }
}
#define SHI_SPEC 4096
#ifndef SHI_INNER_SWEEPS
#define SHI_INNER_SWEEPS 1
#endif
#ifndef SHI_TILED_SWEEPS
#define SHI_TILED_SWEEPS 5
#endif
#ifndef SHI_LIST_SWEEPS
#define SHI_LIST_SWEEPS 8
#endif
#define SHI_TAIL_SWEEPS 0  // the one-workgroup tail (<= 40 sweeps for the last few hundred pixels) cost 160 us of device time per image to spare the
                           // host resolver a few hundred survivors: with the device as the bound of the pipeline it is off (SFMX_SHI_SWEEPS=5,8,40)

static int shi_enqueue(sfmx_ctx* c, const sfmx_pyramid* p, double quality, int min_dist, int cap) {
  unsigned long long* d_max = c->d[1].as<unsigned long long>();
  int* d_rows = c->d[2].as<int>();                // [h] survivors per row -> offsets
  int* d_rows_all = d_rows + p->h + 1;            // [h] candidates per row -> offsets
  int* d_changed = d_rows_all + p->h + 1;
  uint8_t* d_flag = c->d[5].as<uint8_t>();
  uint32_t* d_xy = c->d[3].as<uint32_t>();
  int32_t* d_full = reinterpret_cast<int32_t*>(d_xy + cap);
  int rc = launch_score(c, p, c->d[0].as<double>(), d_max);
  if (rc) return rc;
  dim3 g((p->w + 63) / 64, (p->h + 3) / 4);
  prof_begin(c, KID_SHI_FIXPOINT);  // init + dense sweeps + work-list sweeps + tail + compaction
  // Schedule of the fixpoint: the tile-resident kernel (k_shi_tile), three passes -- 142 us and 3 launches per 640x480 image against
  // 170 us and ~20 launches for the sweep schedule below, 213 against 460 us at 1920x1080 (profiles/r03_shi_tile_probe.txt).  At VGA
  // the interleaved A/B of the end of round 3 reads 27.6 against 28.0 ms per 47-frame pass (profiles/r03_ab_inproc_shi2.txt; an
  // earlier comparison between separate runs had it 4 % slower, when other lanes bounded the pass).  SFMX_SHI_MODE=tile[,passes[,
  // rounds]] | sweeps overrides; any schedule is exact (the host resolver finishes whatever the device leaves undecided).
  int tile_rounds = 64;
  const int tile_passes = shi_tile_passes(p->w, p->h, &tile_rounds);
  if (tile_passes > 0) {
    const size_t lds = shi_tile_lds(min_dist, c->wl_ntaps);
    for (int pass = 0; pass < tile_passes; ++pass) {
      const int offx = (pass & 1) ? SHT_TW / 2 : 0, offy = (pass & 1) ? SHT_TH / 2 : 0;  // odd passes: the tile grid shifted by half a tile
      dim3 gt((p->w + offx + SHT_TW - 1) / SHT_TW, (p->h + offy + SHT_TH - 1) / SHT_TH);
      k_shi_tile<<<gt, SHT_THREADS, lds, c->stream>>>(c->d[0].as<double>(), p->w, p->h, min_dist, d_flag, d_max, quality, pass == 0 ? 1 : 0, offx, offy,
                                                     c->wl[3].as<int8_t>(), c->wl_ntaps, tile_rounds);
    }
  } else {
  k_shi_init<<<g, 256, 0, c->stream>>>(c->d[0].as<double>(), p->w, p->h, d_max, quality, d_flag, c->wl[2].as<int>(), SHI_LIST_SWEEPS + 2);
  // Sweep 1 over all pixels (LDS tiles), then a fixed number of work-list sweeps.  The fixpoint is normally
  // reached after ~30 sweeps; later sweeps see an empty list and cost ~2 us, and stopping before the
  // fixpoint is always safe (undecided pixels simply travel to the host).
  {
    const int r = min_dist - 1;
    const size_t shm = (size_t)(SR_TX + 2 * r) * (SR_TY + 2 * r) * 9 + 16;
    dim3 gt((p->w + SR_TX - 1) / SR_TX, (p->h + SR_TY - 1) / SR_TY);
    // SFMX_SHI_SWEEPS="tiled,list,tail" overrides the schedule (A/B; any schedule is exact: undecided pixels travel to the host)
    static int tiled_sweeps = SHI_TILED_SWEEPS, list_sweeps = SHI_LIST_SWEEPS, tail_sweeps = SHI_TAIL_SWEEPS;
    static const int inner_sweeps = getenv("SFMX_SHI_INNER") ? std::max(1, std::min(8, atoi(getenv("SFMX_SHI_INNER")))) : SHI_INNER_SWEEPS;
    static const bool parsed = [] {
      if (const char* e = getenv("SFMX_SHI_SWEEPS")) {
        int a = -1, b = -1, t = -1;
        if (sscanf(e, "%d,%d,%d", &a, &b, &t) == 3 && a >= 1 && a <= 64 && b >= 0 && b <= SHI_LIST_SWEEPS && t >= 0 && t <= 1000) {
          tiled_sweeps = a; list_sweeps = b; tail_sweeps = t;
        }
      }
      return true;
    }();
    (void)parsed;
    for (int k = 0; k < tiled_sweeps; ++k)  // dense phase: LDS-tiled sweeps over all pixels
      k_shi_round<<<gt, 256, shm, c->stream>>>(c->d[0].as<double>(), p->w, p->h, min_dist, d_flag, d_changed, inner_sweeps);
    const int npx = p->w * p->h;
    uint32_t* list0 = c->wl[0].as<uint32_t>();
    uint32_t* list1 = c->wl[1].as<uint32_t>();
    int* counts = c->wl[2].as<int>();
    if (list_sweeps > 0 || tail_sweeps > 0) k_shi_list_build<<<(npx + 255) / 256, 256, 0, c->stream>>>(d_flag, npx, list0, counts);
    for (int k = 0; k < list_sweeps; ++k)   // sparse phase: work-list sweeps
      k_shi_list_sweep<<<k < 3 ? 1024 : 256, 256, 0, c->stream>>>(c->d[0].as<double>(), p->w, p->h, d_flag, c->wl[3].as<int8_t>(), c->wl_ntaps,
                                                                  (k & 1) ? list1 : list0, counts + k, (k & 1) ? list0 : list1, counts + k + 1);
    // tail: the remaining sweeps inside one workgroup
    if (tail_sweeps > 0)
      k_shi_list_tail<<<1, 1024, 0, c->stream>>>(c->d[0].as<double>(), p->w, p->h, d_flag, c->wl[3].as<int8_t>(), c->wl_ntaps,
                                                 (list_sweeps & 1) ? list1 : list0, (list_sweeps & 1) ? list0 : list1, counts + list_sweeps, tail_sweeps);
  }
  }
  k_flag_row_count<<<p->h, 256, 0, c->stream>>>(d_flag, p->w, d_rows, d_rows_all);
  // header in d[1]: [0] max score bits (8 B) | [8] #survivors (4 B) | [12] #candidates (4 B)
  int* d_tot = reinterpret_cast<int*>(c->d[1].as<char>() + 8);
  k_row_scan<<<1, 64, 0, c->stream>>>(d_rows, p->h, d_tot);
  k_row_scan<<<1, 64, 0, c->stream>>>(d_rows_all, p->h, d_tot + 1);
  // the 16-byte header plus the first SPEC survivors (there are ~1.5-2 k per VGA frame) go straight into pinned memory;
  // a download happens only if there are more
  const int SPEC = cap < SHI_SPEC ? cap : SHI_SPEC;
  k_flag_row_write<<<p->h, 64, 0, c->stream>>>(c->d[0].as<double>(), d_flag, p->w, d_rows, d_rows_all, cap, d_xy, c->d[4].as<double>(), d_full,
                                          c->d[6].as<double2>(), c->d[1].as<unsigned long long>(), c->h[2].as<char>(), SPEC);
  prof_end(c);
  SFMX_HIP(c, hipGetLastError());
  return SFMX_OK;
}

int sfmx_shi_tomasi_candidates_pruned(sfmx_ctx* c, const sfmx_pyramid* p, double quality, int min_dist, int cap, uint32_t* cand_xy,
                                      double* cand_score, int32_t* cand_full_index, int* n_out, int* n_total_out, double* max_out) {
  SFMX_REQUIRE(c, c && p && cand_xy && cand_score && n_out && cap > 0 && p->w < 32768 && p->h < 32768 && min_dist >= 1 && min_dist <= SR_MAXR + 1);
  const size_t n = (size_t)p->w * p->h;
  c->resident_points = 0;
  c->shi_full_count = 0;
  if (c->shi_keys_in_flight) {
    SFMX_HIP(c, hipStreamSynchronize(c->copy_stream));
    c->shi_keys_in_flight = false;
  }
  const int SPEC = cap < SHI_SPEC ? cap : SHI_SPEC;
  SFMX_HIP(c, c->d[0].ensure(n * 8));
  SFMX_HIP(c, c->d[1].ensure(64));
  SFMX_HIP(c, c->d[2].ensure((size_t)(2 * p->h + 8) * 4 + 64));
  SFMX_HIP(c, c->d[3].ensure((size_t)cap * 8));   // xy (u32) + full index (i32)
  SFMX_HIP(c, c->d[4].ensure((size_t)cap * 8));
  SFMX_HIP(c, c->d[5].ensure(n + 64));
  SFMX_HIP(c, c->d[6].ensure(n * 16));            // sort keys {score, id, mark} of all candidates, row-major
  SFMX_HIP(c, c->h[2].ensure(64 + (size_t)SPEC * 16));
  SFMX_HIP(c, c->wl[0].ensure(n * 4));
  SFMX_HIP(c, c->wl[1].ensure(n * 4));
  SFMX_HIP(c, c->wl[2].ensure((SHI_LIST_SWEEPS + 2) * sizeof(int)));
  if (c->wl_md != min_dist) {  // offsets of the open disc dx^2 + dy^2 < min_dist^2 (centre excluded)
    std::vector<int8_t> taps;
    for (int dy = -(min_dist - 1); dy <= min_dist - 1; ++dy)
      for (int dx = -(min_dist - 1); dx <= min_dist - 1; ++dx)
        if ((dx || dy) && dx * dx + dy * dy < min_dist * min_dist) { taps.push_back((int8_t)dx); taps.push_back((int8_t)dy); }
    SFMX_HIP(c, c->wl[3].ensure(taps.size() + 16));
    SFMX_HIP(c, hipMemcpyAsync(c->wl[3].p, taps.data(), taps.size(), hipMemcpyHostToDevice, c->stream));
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
    c->wl_ntaps = (int)taps.size() / 2;
    c->wl_md = min_dist;
  }
  uint32_t* d_xy = c->d[3].as<uint32_t>();
  int32_t* d_full = reinterpret_cast<int32_t*>(d_xy + cap);

  if (shi_tile_lds(min_dist, c->wl_ntaps) > 64 * 1024) {  // large discs: the tile kernel needs the dynamic-LDS opt-in (per device, once)
    static std::mutex attr_mu;
    static bool attr_set[64] = {};
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!attr_set[c->device & 63]) {
      SFMX_HIP(c, hipFuncSetAttribute((const void*)k_shi_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      attr_set[c->device & 63] = true;
    }
  }
  static const bool no_graph = getenv("SFMX_NO_GRAPH") != nullptr;
  ShiGraphKey key;
  memset(&key, 0, sizeof key);
  key.img = p->base; key.d0 = c->d[0].p; key.d1 = c->d[1].p; key.d2 = c->d[2].p; key.d3 = c->d[3].p; key.d4 = c->d[4].p;
  key.d5 = c->d[5].p; key.d6 = c->d[6].p; key.pin = c->h[2].p; key.w0 = c->wl[0].p; key.w1 = c->wl[1].p; key.w2 = c->wl[2].p; key.w3 = c->wl[3].p; key.w = p->w; key.h = p->h; key.md = min_dist; key.cap = cap;
  key.tile_passes = shi_tile_passes(p->w, p->h, &key.tile_rounds);
  key.quality = quality;
  bool launched = false;
  if (!no_graph && !c->timing) {
    auto& cache = shi_graphs(c);
    hipGraphExec_t exec = nullptr;
    for (auto& g : cache) if (g.key == key) exec = g.exec;
    if (!exec) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = shi_enqueue(c, p, quality, min_dist, cap);
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (rc == SFMX_OK && e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          if (cache.size() >= 8) {  // bounded cache: drop the oldest executable
            (void)hipGraphExecDestroy(cache.front().exec);
            cache.erase(cache.begin());
          }
          cache.push_back(ShiGraph{key, exec});
        } else {
          exec = nullptr;
        }
        if (graph) (void)hipGraphDestroy(graph);
      }
    }
    if (exec) {
      SFMX_HIP(c, hipGraphLaunch(exec, c->stream));
      launched = true;
    }
  }
  if (!launched) {
    const int rc = shi_enqueue(c, p, quality, min_dist, cap);
    if (rc) return rc;
  }
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  KernelTimer(c).collect();
  const size_t o_xy = 64, o_s = o_xy + (size_t)SPEC * 4, o_full = o_s + (size_t)SPEC * 8;
  const char* pin = c->h[2].as<char>();
  double mx;
  int tot, tot_all;
  memcpy(&mx, pin, 8);
  memcpy(&tot, pin + 8, 4);
  memcpy(&tot_all, pin + 12, 4);
  const int m = tot < cap ? tot : cap;
  if (m <= SPEC) {
    memcpy(cand_xy, pin + o_xy, (size_t)m * 4);
    memcpy(cand_score, pin + o_s, (size_t)m * 8);
    if (cand_full_index) memcpy(cand_full_index, pin + o_full, (size_t)m * 4);
  } else {
    SFMX_HIP(c, hipMemcpyAsync(cand_xy, d_xy, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    SFMX_HIP(c, hipMemcpyAsync(cand_score, c->d[4].p, (size_t)m * 8, hipMemcpyDeviceToHost, c->stream));
    if (cand_full_index) SFMX_HIP(c, hipMemcpyAsync(cand_full_index, d_full, (size_t)m * 4, hipMemcpyDeviceToHost, c->stream));
    SFMX_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->shi_full_count = tot_all;
  *n_out = tot;
  if (n_total_out) *n_total_out = tot_all;
  if (max_out) *max_out = mx;
  // Speculative download of the full key list (2 MB per VGA frame) on the copy stream: most frames need it for
  // the tie-order replay, and it overlaps with the host's sort + walk over the survivors.
  if (tot_all > 0) {
    SFMX_HIP(c, c->h[3].ensure((size_t)tot_all * 16));
    SFMX_HIP(c, hipMemcpyAsync(c->h[3].p, c->d[6].p, (size_t)tot_all * 16, hipMemcpyDeviceToHost, c->copy_stream));
    c->shi_keys_in_flight = true;
  }
  return SFMX_OK;
}

// scores of ALL candidates (row-major) of the preceding sfmx_shi_tomasi_candidates_pruned call, still in HBM
int sfmx_shi_tomasi_fetch_all_keys(sfmx_ctx* c, int n_total, void** keys_out) {
  SFMX_REQUIRE(c, c && keys_out && n_total > 0 && n_total == c->shi_full_count);
  if (!c->shi_keys_in_flight) {
    SFMX_HIP(c, c->h[3].ensure((size_t)n_total * 16));
    SFMX_HIP(c, hipMemcpyAsync(c->h[3].p, c->d[6].p, (size_t)n_total * 16, hipMemcpyDeviceToHost, c->copy_stream));
  }
  SFMX_HIP(c, hipStreamSynchronize(c->copy_stream));
  c->shi_keys_in_flight = false;
  *keys_out = c->h[3].p;  // pinned host memory owned by the context; valid until the next Shi-Tomasi call
  return SFMX_OK;
}

}  // extern "C"
