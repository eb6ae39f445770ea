// fake_sfmx.cpp -- TEST INFRASTRUCTURE, never shipped: a CPU stand-in for libsfmx.so (include/sfmx.h) whose arithmetic is
// the oracle's (oracle/sfm_oracle.cpp, linked as liborc).  It exists so that the HOST runtime (csrc/host: seven lanes, thread
// pool, arena, polled joins) can run in a container without a GPU -- built with -fsanitize=thread / address,undefined -- and so
// that the multi-process sharded mode (BA points + RANSAC hypotheses over N ranks, two communicators) can be executed with
// real processes: sfmx_comm_* is an all-reduce through a POSIX shared-memory segment instead of RCCL.
// The product (structure-from-motion-3d-reconstruction_amd/_build) never links or loads this file.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/sfmx.h"

typedef unsigned char u8;
extern "C" {  // the oracle's flat C interface (oracle/sfm_oracle.cpp)
void orc_downsample2(const u8* px, int w, int h, u8* out);
void orc_shi_score(const u8* px, int w, int h, double* score);
void orc_klt_track(const u8* ia, const u8* ib, int w, int h, int levels, int radius, int iters, double fb_thresh, int n, const double* xy_in,
                   double* xy_fwd, double* xy_back, u8* keep);
void orc_ransac_hypotheses(const double* xn, const double* yn, const int* idx8, int H, double* E_out);
void orc_ransac_counts(const double* xn, const double* yn, int n, const double* E, int H, double thr, int* counts);
double orc_sampson_err(const double* E9, double x, double y, double xp, double yp);
void orc_ba_build(const double* poses_wc, int W, const double* X, int P, const int* obs_ptr, const int* obs_li, const double* obs_uv, double fx,
                  double fy, double cx, double cy, double huber, double lambda, int damp, double* S, double* b);
int orc_solve_gauss(const double* A, const double* b, int n, double* x);
}

struct sfmx_ctx {
  int device = 0;
  std::string err;
  bool timing = false;
  std::vector<double> xi, xj;  // correspondences left by the last sfmx_ransac_score_ex (sfmx_sampson_mask reuses them)
  std::vector<double> keys;    // {score, (id, mark)} records of all candidates of the last pruned Shi-Tomasi call
};
struct sfmx_pyramid {
  int w = 0, h = 0, levels = 0;
  std::vector<std::vector<u8>> px;
  std::vector<int> lw, lh;
};
struct sfmx_ba_problem {
  int W = 0, P = 0;
  std::vector<double> X, uv;
  std::vector<int32_t> ptr, li;
};

namespace {
int fail(sfmx_ctx* c, int st, const char* what) {
  if (c) c->err = what;
  return st;
}
void build_levels(sfmx_pyramid* p) {
  for (int l = 1; l < p->levels; l++) {
    p->px[(size_t)l].assign((size_t)std::max(1, p->lw[(size_t)l] * p->lh[(size_t)l]), 0);
    if (p->lw[(size_t)l] > 0 && p->lh[(size_t)l] > 0)
      orc_downsample2(p->px[(size_t)l - 1].data(), p->lw[(size_t)l - 1], p->lh[(size_t)l - 1], p->px[(size_t)l].data());
  }
}
}  // namespace

extern "C" {

int sfmx_ctx_create_prio(int device_id, int, sfmx_ctx** out) {
  if (!out) return SFMX_ERR_INVALID;
  *out = new sfmx_ctx;
  (*out)->device = device_id;
  return SFMX_OK;
}
int sfmx_ctx_create(int device_id, sfmx_ctx** out) { return sfmx_ctx_create_prio(device_id, 0, out); }
void sfmx_ctx_destroy(sfmx_ctx* c) { delete c; }
const char* sfmx_last_error(const sfmx_ctx* c) { return c ? c->err.c_str() : "null context"; }
int sfmx_sync(sfmx_ctx* c) { return c ? SFMX_OK : SFMX_ERR_INVALID; }
int sfmx_ctx_device(const sfmx_ctx* c) { return c ? c->device : -1; }
int sfmx_ctx_make_current(sfmx_ctx* c) { return c ? SFMX_OK : SFMX_ERR_INVALID; }
void* sfmx_stream(sfmx_ctx*) { return nullptr; }
int sfmx_set_timing(sfmx_ctx* c, int e) { if (!c) return SFMX_ERR_INVALID; c->timing = e != 0; return SFMX_OK; }
int sfmx_get_timing(const sfmx_ctx* c) { return (c && c->timing) ? 1 : 0; }
double sfmx_last_kernel_us(const sfmx_ctx*) { return 0.0; }
int sfmx_kernel_profile(sfmx_ctx*, int, int cap, double* us, uint64_t* calls) {
  for (int i = 0; i < cap && i < 10; i++) { if (us) us[i] = 0; if (calls) calls[i] = 0; }
  return 10;
}
const char* sfmx_kernel_profile_name(int) { return ""; }

int sfmx_pyramid_create(sfmx_ctx* c, int w, int h, int levels, sfmx_pyramid** out) {
  if (!c || !out || w <= 0 || h <= 0 || levels < 1 || levels > 8) return fail(c, SFMX_ERR_INVALID, "pyramid_create");
  sfmx_pyramid* p = new sfmx_pyramid;
  p->w = w; p->h = h; p->levels = levels;
  p->px.resize((size_t)levels);
  int lw = w, lh = h;
  for (int l = 0; l < levels; l++) { p->lw.push_back(lw); p->lh.push_back(lh); lw /= 2; lh /= 2; }
  p->px[0].assign((size_t)w * h, 0);
  build_levels(p);
  *out = p;
  return SFMX_OK;
}
void sfmx_pyramid_destroy(sfmx_ctx*, sfmx_pyramid* p) { delete p; }
int sfmx_pyramid_upload(sfmx_ctx* c, sfmx_pyramid* p, const uint8_t* host) {
  if (!c || !p || !host) return fail(c, SFMX_ERR_INVALID, "pyramid_upload");
  std::memcpy(p->px[0].data(), host, (size_t)p->w * p->h);
  build_levels(p);
  return SFMX_OK;
}
int sfmx_pyramid_set_device(sfmx_ctx* c, sfmx_pyramid* p, const void* dev) { return sfmx_pyramid_upload(c, p, static_cast<const uint8_t*>(dev)); }
int sfmx_pyramid_set_device_async(sfmx_ctx* c, sfmx_pyramid* p, const void* dev, int) { return sfmx_pyramid_upload(c, p, static_cast<const uint8_t*>(dev)); }
int sfmx_pyramid_wait(sfmx_ctx* c, sfmx_pyramid* p) { return (c && p) ? SFMX_OK : SFMX_ERR_INVALID; }
int sfmx_pyramid_fetched_level(sfmx_ctx* c, sfmx_pyramid* p, int, const uint8_t** out) {
  if (!c || !p || !out) return SFMX_ERR_INVALID;
  *out = nullptr;  // nothing is fetched ahead of time here: the caller downloads
  return SFMX_OK;
}
int sfmx_pyramid_download_level(sfmx_ctx* c, const sfmx_pyramid* p, int level, uint8_t* out) {
  if (!c || !p || !out || level < 0 || level >= p->levels) return fail(c, SFMX_ERR_INVALID, "pyramid_download_level");
  std::memcpy(out, p->px[(size_t)level].data(), (size_t)p->lw[(size_t)level] * p->lh[(size_t)level]);
  return SFMX_OK;
}
int sfmx_pyramid_level_size(const sfmx_pyramid* p, int level, int* w, int* h) {
  if (!p || level < 0 || level >= p->levels) return SFMX_ERR_INVALID;
  if (w) *w = p->lw[(size_t)level];
  if (h) *h = p->lh[(size_t)level];
  return SFMX_OK;
}

// every candidate is reported as "undecided": stopping the device fixpoint before it starts is a valid (exact) schedule,
// the host resolver then does the whole pick
int sfmx_shi_tomasi_candidates_pruned(sfmx_ctx* c, const sfmx_pyramid* p, double quality, int, int cap, uint32_t* cand_xy, double* cand_score,
                                      int32_t* cand_full_index, int* n_out, int* n_total_out, double* max_out) {
  if (!c || !p || !cand_xy || !cand_score || !n_out) return fail(c, SFMX_ERR_INVALID, "shi_tomasi_candidates_pruned");
  std::vector<double> s((size_t)p->w * p->h);
  orc_shi_score(p->px[0].data(), p->w, p->h, s.data());
  double mx = 0.0;
  for (double v : s) mx = v > mx ? v : mx;
  const double thr = mx * quality;
  int n = 0;
  c->keys.clear();
  for (int y = 0; y < p->h; y++)
    for (int x = 0; x < p->w; x++) {
      const double v = s[(size_t)y * p->w + x];
      if (!(v >= thr)) continue;
      if (n < cap) {
        cand_xy[n] = (uint32_t)x | ((uint32_t)y << 16);
        cand_score[n] = v;
        if (cand_full_index) cand_full_index[n] = n;
      }
      double rec[2];
      rec[0] = v;
      const uint32_t idm[2] = {(uint32_t)n, 0u};
      std::memcpy(&rec[1], idm, 8);
      c->keys.push_back(rec[0]);
      c->keys.push_back(rec[1]);
      n++;
    }
  *n_out = n;
  if (n_total_out) *n_total_out = n;
  if (max_out) *max_out = mx;
  return SFMX_OK;
}
int sfmx_shi_tomasi_fetch_all_keys(sfmx_ctx* c, int n_total, void** keys_out) {
  if (!c || !keys_out || (size_t)n_total * 2 != c->keys.size()) return fail(c, SFMX_ERR_INVALID, "fetch_all_keys");
  *keys_out = c->keys.data();
  return SFMX_OK;
}
int sfmx_shi_tomasi_candidates(sfmx_ctx* c, const sfmx_pyramid* p, double quality, int cap, uint32_t* cand_xy, double* cand_score, int* n_out,
                               double* max_out) {
  int tot = 0;
  return sfmx_shi_tomasi_candidates_pruned(c, p, quality, 1, cap, cand_xy, cand_score, nullptr, n_out, &tot, max_out);
}

int sfmx_klt_track(sfmx_ctx* c, const sfmx_pyramid* a, const sfmx_pyramid* b, const double* xy_in, int n, const sfmx_klt_cfg* cfg, double* xy_fwd,
                   double* xy_back, uint8_t* keep, uint64_t* n_steps) {
  if (!c || !a || !b || !cfg || !xy_fwd || !keep || n < 0) return fail(c, SFMX_ERR_INVALID, "klt_track");
  if (n_steps) *n_steps = 0;
  if (n == 0) return SFMX_OK;
  std::vector<double> back((size_t)2 * n);
  orc_klt_track(a->px[0].data(), b->px[0].data(), a->w, a->h, cfg->levels, cfg->win_radius, cfg->iters, cfg->fb_thresh, n, xy_in, xy_fwd,
                back.data(), keep);
  if (xy_back) std::memcpy(xy_back, back.data(), back.size() * 8);
  return SFMX_OK;
}

int sfmx_ransac_score_ex(sfmx_ctx* c, const double* xi, const double* xj, int n, const int32_t* idx8, int H, double thr, int32_t* counts,
                         int32_t* lo, int32_t* hi, uint8_t* flags, double* cond, int32_t* best_iter, int32_t* best_count, double* E_out) {
  if (!c || !xi || !xj || !idx8 || n < 8 || H <= 0 || !best_iter || !best_count) return fail(c, SFMX_ERR_INVALID, "ransac_score_ex");
  std::vector<double> E((size_t)H * 9);
  orc_ransac_hypotheses(xi, xj, idx8, H, E.data());
  std::vector<int> cnt((size_t)H);
  orc_ransac_counts(xi, xj, n, E.data(), H, thr, cnt.data());
  int bi = 0, bc = cnt[0];
  for (int h = 0; h < H; h++) {
    if (counts) counts[h] = cnt[(size_t)h];
    if (lo) lo[h] = cnt[(size_t)h];  // every hypothesis is the exact host one
    if (hi) hi[h] = cnt[(size_t)h];
    if (flags) flags[h] = 1;
    if (cond) cond[h] = INFINITY;
    if (cnt[(size_t)h] > bc) { bc = cnt[(size_t)h]; bi = h; }
  }
  *best_iter = bi;
  *best_count = bc;
  if (E_out) std::memcpy(E_out, E.data(), E.size() * 8);
  c->xi.assign(xi, xi + (size_t)2 * n);
  c->xj.assign(xj, xj + (size_t)2 * n);
  return SFMX_OK;
}
int sfmx_sampson_mask(sfmx_ctx* c, const double* xi, const double* xj, int n, const double* E9, double thr, uint8_t* mask, int32_t* count) {
  if (!c || !E9 || !mask || n <= 0) return fail(c, SFMX_ERR_INVALID, "sampson_mask");
  if (!xi || !xj) {
    if ((int)c->xi.size() != 2 * n) return fail(c, SFMX_ERR_INVALID, "sampson_mask: no resident points");
    xi = c->xi.data();
    xj = c->xj.data();
  }
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    mask[i] = orc_sampson_err(E9, xi[2 * i], xi[2 * i + 1], xj[2 * i], xj[2 * i + 1]) < thr ? 1 : 0;
    cnt += mask[i];
  }
  if (count) *count = cnt;
  return SFMX_OK;
}

int sfmx_ba_reset(sfmx_ctx* c, sfmx_ba_problem* q, int W, int P, const double* X, const int32_t* ptr, const int32_t* li, const double* uv) {
  if (!c || !q || W < 1 || P < 1 || !X || !ptr || !li || !uv) return fail(c, SFMX_ERR_INVALID, "ba_reset");
  q->W = W; q->P = P;
  q->X.assign(X, X + (size_t)3 * P);
  q->ptr.assign(ptr, ptr + P + 1);
  q->li.assign(li, li + ptr[P]);
  q->uv.assign(uv, uv + (size_t)2 * ptr[P]);
  return SFMX_OK;
}
int sfmx_ba_create(sfmx_ctx* c, int W, int P, const double* X, const int32_t* ptr, const int32_t* li, const double* uv, sfmx_ba_problem** out) {
  if (!out) return SFMX_ERR_INVALID;
  sfmx_ba_problem* q = new sfmx_ba_problem;
  const int rc = sfmx_ba_reset(c, q, W, P, X, ptr, li, uv);
  if (rc) { delete q; return rc; }
  *out = q;
  return SFMX_OK;
}
void sfmx_ba_destroy(sfmx_ctx*, sfmx_ba_problem* q) { delete q; }
static int ba_solve(sfmx_ctx* c, int D, std::vector<double>& S, std::vector<double>& b, double* dx) {
  (void)c;
  return orc_solve_gauss(S.data(), b.data(), D, dx) ? SFMX_ERR_SINGULAR : SFMX_OK;
}
int sfmx_ba_begin(sfmx_ctx* c, sfmx_ba_problem* q, int, double, double, double, double, double, double) { return (c && q) ? SFMX_OK : SFMX_ERR_INVALID; }
int sfmx_ba_end(sfmx_ctx* c, sfmx_ba_problem* q) { return (c && q) ? SFMX_OK : SFMX_ERR_INVALID; }
int sfmx_ba_step(sfmx_ctx* c, sfmx_ba_problem* q, const double* poses, double fx, double fy, double cx, double cy, double huber, double lambda,
                 double* dx) {
  if (!c || !q || !poses || !dx) return fail(c, SFMX_ERR_INVALID, "ba_step");
  const int D = 6 * q->W;
  std::vector<double> S((size_t)D * D), b((size_t)D);
  orc_ba_build(poses, q->W, q->X.data(), q->P, q->ptr.data(), q->li.data(), q->uv.data(), fx, fy, cx, cy, huber, lambda, 1, S.data(), b.data());
  return ba_solve(c, D, S, b, dx);
}

// ---- communicators: all-reduce through a shared-memory segment (one segment per communicator) -----------------------------
struct ShmHeader {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
};
}  // extern "C"
struct sfmx_comm {
  int rank = 0, world = 1, device = 0;
  std::string name;
  ShmHeader* hdr = nullptr;
  unsigned char* slots = nullptr;  // [world][SLOT]
  size_t bytes = 0;
};
namespace {
constexpr size_t kSlot = 64 * 1024;  // the largest payload of the path is D*D + D doubles at D = 60: 29 280 bytes
bool barrier(sfmx_comm* m) {
  ShmHeader* h = m->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == m->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_acq_rel);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->generation.load(std::memory_order_acquire) == gen) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;  // a peer never came: report, do not hang
  }
  return true;
}
template <class T, class F>
int allreduce(sfmx_ctx* c, sfmx_comm* m, T* data, size_t n, F op) {
  if (!m || m->world <= 1 || n == 0) return SFMX_OK;
  if (n * sizeof(T) > kSlot) return fail(c, SFMX_ERR_INVALID, "fake all-reduce: payload too large");
  std::memcpy(m->slots + (size_t)m->rank * kSlot, data, n * sizeof(T));
  if (!barrier(m)) return fail(c, SFMX_ERR_HIP, "fake all-reduce: timed out waiting for the other ranks");
  std::vector<T> acc(n);
  std::memcpy(acc.data(), m->slots, n * sizeof(T));
  for (int r = 1; r < m->world; r++) {  // rank order
    const T* o = reinterpret_cast<const T*>(m->slots + (size_t)r * kSlot);
    for (size_t i = 0; i < n; i++) acc[i] = op(acc[i], o[i]);
  }
  if (!barrier(m)) return fail(c, SFMX_ERR_HIP, "fake all-reduce: timed out waiting for the other ranks");
  std::memcpy(data, acc.data(), n * sizeof(T));
  return SFMX_OK;
}
}  // namespace
extern "C" {
int sfmx_comm_get_unique_id(void* id_out) {
  if (!id_out) return SFMX_ERR_INVALID;
  std::memset(id_out, 0, SFMX_COMM_ID_BYTES);
  static std::atomic<unsigned> serial{0};
  std::snprintf(static_cast<char*>(id_out), SFMX_COMM_ID_BYTES, "/sfmx_fake_%d_%u_%lld", (int)getpid(), serial.fetch_add(1),
                (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return SFMX_OK;
}
int sfmx_comm_create(int device, const void* id, int rank, int world, sfmx_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world) return SFMX_ERR_INVALID;
  sfmx_comm* m = new sfmx_comm;
  m->rank = rank; m->world = world; m->device = device;
  if (world > 1) {
    if (!id) { delete m; return SFMX_ERR_INVALID; }
    m->name = static_cast<const char*>(id);
    m->bytes = 4096 + (size_t)world * kSlot;
    const int fd = shm_open(m->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)m->bytes) != 0) { if (fd >= 0) close(fd); delete m; return SFMX_ERR_HIP; }
    void* p = mmap(nullptr, m->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);  // a fresh segment is zero-filled
    close(fd);
    if (p == MAP_FAILED) { delete m; return SFMX_ERR_HIP; }
    m->hdr = static_cast<ShmHeader*>(p);
    m->slots = static_cast<unsigned char*>(p) + 4096;
    m->hdr->attached.fetch_add(1);
  }
  *out = m;
  return SFMX_OK;
}
void sfmx_comm_destroy(sfmx_comm* m) {
  if (!m) return;
  if (m->hdr) {
    const bool last = m->hdr->attached.fetch_sub(1) == 1;
    munmap(m->hdr, m->bytes);
    if (last) shm_unlink(m->name.c_str());
  }
  delete m;
}
int sfmx_comm_rank(const sfmx_comm* m) { return m ? m->rank : 0; }
int sfmx_comm_world(const sfmx_comm* m) { return m ? m->world : 1; }
void sfmx_shard_range(int n, int rank, int world, int* lo, int* hi) {
  if (world < 1) world = 1;
  const int base = n / world, extra = n % world;
  const int l = rank * base + (rank < extra ? rank : extra);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (rank < extra ? 1 : 0);
}
int sfmx_comm_allreduce_f64(sfmx_ctx* c, sfmx_comm* m, double* v, int n, int op) {
  if (op == 0) return allreduce(c, m, v, (size_t)n, [](double a, double b) { return a + b; });
  return allreduce(c, m, v, (size_t)n, [](double a, double b) { return a > b ? a : b; });
}
int sfmx_comm_allreduce_u64_max(sfmx_ctx* c, sfmx_comm* m, uint64_t* v, int n) {
  return allreduce(c, m, v, (size_t)n, [](uint64_t a, uint64_t b) { return a > b ? a : b; });
}
int sfmx_ba_step_sharded(sfmx_ctx* c, sfmx_comm* m, sfmx_ba_problem* q, const double* poses, double fx, double fy, double cx, double cy,
                         double huber, double lambda, double* dx) {
  if (!c || !q || !poses || !dx) return fail(c, SFMX_ERR_INVALID, "ba_step_sharded");
  const int D = 6 * q->W;
  std::vector<double> Sb((size_t)D * D + D);
  orc_ba_build(poses, q->W, q->X.data(), q->P, q->ptr.data(), q->li.data(), q->uv.data(), fx, fy, cx, cy, huber, 0.0, 0, Sb.data(),
               Sb.data() + (size_t)D * D);
  const int rc = allreduce(c, m, Sb.data(), Sb.size(), [](double a, double b) { return a + b; });
  if (rc) return rc;
  std::vector<double> S(Sb.begin(), Sb.begin() + (size_t)D * D), b(Sb.begin() + (size_t)D * D, Sb.end());
  for (int i = 0; i < D; i++) {  // T:1064-1071
    S[(size_t)i * D + i] += lambda;
    if (i < 6) { S[(size_t)i * D + i] += 1e9; b[(size_t)i] = 0.0; }
  }
  return ba_solve(c, D, S, b, dx);
}

// element sharding: this rank's slice of S | b (blocks of 16 elements, as the product's reduction kernel shards them), +0.0 elsewhere
int sfmx_ba_step_sharded_elements(sfmx_ctx* c, sfmx_comm* m, sfmx_ba_problem* q, const double* poses, double fx, double fy, double cx, double cy,
                                  double huber, double lambda, double* dx) {
  if (!c || !q || !poses || !dx) return fail(c, SFMX_ERR_INVALID, "ba_step_sharded_elements");
  const int D = 6 * q->W, NE = D * D + D;
  std::vector<double> Sb((size_t)NE);
  orc_ba_build(poses, q->W, q->X.data(), q->P, q->ptr.data(), q->li.data(), q->uv.data(), fx, fy, cx, cy, huber, 0.0, 0, Sb.data(),
               Sb.data() + (size_t)D * D);
  if (m && m->world > 1) {
    const int nblk = (NE + 15) / 16;
    int lo = 0, hi = nblk;
    sfmx_shard_range(nblk, m->rank, m->world, &lo, &hi);
    for (int e = 0; e < NE; e++)
      if (e < lo * 16 || e >= hi * 16) Sb[(size_t)e] = 0.0;
    const int rc = allreduce(c, m, Sb.data(), Sb.size(), [](double a, double b) { return a + b; });
    if (rc) return rc;
  }
  std::vector<double> S(Sb.begin(), Sb.begin() + (size_t)D * D), b(Sb.begin() + (size_t)D * D, Sb.end());
  for (int i = 0; i < D; i++) {  // T:1064-1071
    S[(size_t)i * D + i] += lambda;
    if (i < 6) { S[(size_t)i * D + i] += 1e9; b[(size_t)i] = 0.0; }
  }
  return ba_solve(c, D, S, b, dx);
}

int sfmx_solve_dense(sfmx_ctx* c, const double* A, const double* b, int n, double* x) {
  if (!c || !A || !b || !x || n < 1) return fail(c, SFMX_ERR_INVALID, "solve_dense");
  return orc_solve_gauss(A, b, n, x) ? SFMX_ERR_SINGULAR : SFMX_OK;
}
// the structured system assembled densely (test sizes only): L (x) I_3
int sfmx_posegraph_solve(sfmx_ctx* c, int n, const int32_t* ij, const double* v, int m, const double* g3, double* x3) {
  if (!c || n < 1 || !ij || !v || !g3 || !x3) return fail(c, SFMX_ERR_INVALID, "posegraph_solve");
  std::vector<double> L((size_t)n * n, 0.0), rhs((size_t)n), x((size_t)n);
  for (int k = 0; k < m; k++) {
    L[(size_t)ij[2 * k] * n + ij[2 * k + 1]] = v[k];
    L[(size_t)ij[2 * k + 1] * n + ij[2 * k]] = v[k];
  }
  for (int d = 0; d < 3; d++) {
    for (int i = 0; i < n; i++) rhs[(size_t)i] = g3[3 * i + d];
    if (orc_solve_gauss(L.data(), rhs.data(), n, x.data())) return SFMX_ERR_SINGULAR;
    for (int i = 0; i < n; i++) x3[3 * i + d] = x[(size_t)i];
  }
  return SFMX_OK;
}

}  // extern "C"
