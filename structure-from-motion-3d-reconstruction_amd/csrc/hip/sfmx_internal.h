// sfmx_internal.h — context, buffers and error plumbing behind include/sfmx.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../../include/sfmx.h"
#include "sfmx_math.h"

#define SFMX_MAX_LEVELS 8

// Grow-only device / pinned slabs.  Growing never frees: hipFree / hipHostFree wait for EVERY stream of the device, and a
// lane that waits there while another lane's RCCL collective is in flight (blocked on a peer whose own lane waits the same way)
// is a deadlock across ranks.  The outgrown block is parked and released with the buffer (growth is geometric, so the parked
// blocks add up to less than four times the final size).
#include <vector>
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  std::vector<void*> parked;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) parked.push_back(p);
    p = nullptr;
    cap = 0;
    size_t want = n < 4096 ? 4096 : n + n / 4;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    for (void* q : parked) (void)hipFree(q);
    parked.clear();
    p = nullptr;
    cap = 0;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Pinned host memory that kernels read and write directly (track positions, BA poses, polled result blocks): fine-grained
// coherent and mapped, requested explicitly rather than through the runtime's default for flags == 0.
struct PinBuf {
  void* p = nullptr;
  size_t cap = 0;
  std::vector<void*> parked;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) parked.push_back(p);
    p = nullptr;
    cap = 0;
    size_t want = n < 4096 ? 4096 : n + n / 4;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    for (void* q : parked) (void)hipHostFree(q);
    parked.clear();
    p = nullptr;
    cap = 0;
  }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// per-kernel profile (only while ctx->timing): accumulated GPU time and launch count of the hot kernels, measured
// with HIP events recorded on the context's own stream around each launch
enum SfmxKid {
  KID_KLT = 0, KID_HYPOTHESES, KID_SCORE, KID_BA_POINTS, KID_BA_EXPAND, KID_BA_REDUCE, KID_SOLVE, KID_SHI_SCORE, KID_SHI_FIXPOINT,
  KID_PYRAMID, KID_COUNT
};
#define SFMX_PROF_PAIRS 12

struct sfmx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // speculative D2H that must not delay the compute stream
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false;
  double last_us = 0.0;
  hipEvent_t pev[SFMX_PROF_PAIRS][2] = {};
  int prof_kid[SFMX_PROF_PAIRS] = {};
  int prof_n = 0;
  bool prof_open = false;
  double kus[KID_COUNT] = {};
  unsigned long long kcalls[KID_COUNT] = {};
  std::string err;
  // reusable staging: a few independent device / pinned slabs
  DevBuf d[8];
  PinBuf h[5];
  unsigned long long klt_slow_steps = 0;  // lk_steps of the last sfmx_klt_track call that took the per-pixel path
  int resident_points = 0;  // #correspondences left in d[0]/d[1] by the last RANSAC call
  int shi_full_count = 0;   // #candidate scores left in d[6] by the last pruned Shi-Tomasi call
  bool shi_keys_in_flight = false;
  unsigned long long ba_seq = 0;      // sequence number of the last BA step result published into pinned memory
  bool ba_upload_in_flight = false;  // sfmx_ba_reset's transfer out of h[4] has not been waited for yet
  DevBuf wl[4];             // Shi-Tomasi work lists (2), sweep counters, disc offset table
  int wl_md = 0, wl_ntaps = 0;
};

struct sfmx_comm {
  void* nccl = nullptr;  // ncclComm_t; null when world == 1
  int rank = 0, world = 1, device = 0;
};
// comm.hip: in-place all-reduce of device memory on the context's stream (dtype 0 = f64, 1 = u64; op 0 = sum, 1 = max)
int sfmx_comm_allreduce_dev(sfmx_ctx* ctx, sfmx_comm* comm, void* dev, size_t count, int dtype, int op);

struct sfmx_pyramid {
  int w = 0, h = 0, levels = 0;
  uint8_t* base = nullptr;
  size_t off[SFMX_MAX_LEVELS] = {0};
  int lw[SFMX_MAX_LEVELS] = {0}, lh[SFMX_MAX_LEVELS] = {0};
  size_t bytes = 0;
  // sfmx_pyramid_set_device_async: built on the context's second stream; `ready` orders later users behind it
  hipEvent_t ready = nullptr;
  bool ready_pending = false;
  int fetched_level = -1;     // level whose pixels were copied to `fetched` (pinned host memory) by the same call
  PinBuf fetched;
};

// by-value kernel argument describing one pyramid
struct PyrDesc {
  const uint8_t* px[SFMX_MAX_LEVELS];
  int w[SFMX_MAX_LEVELS];
  int h[SFMX_MAX_LEVELS];
  int levels;
};
static inline PyrDesc make_desc(const sfmx_pyramid* p) {
  PyrDesc d;
  for (int l = 0; l < SFMX_MAX_LEVELS; l++) {
    d.px[l] = (l < p->levels) ? p->base + p->off[l] : nullptr;
    d.w[l] = (l < p->levels) ? p->lw[l] : 0;
    d.h[l] = (l < p->levels) ? p->lh[l] : 0;
  }
  d.levels = p->levels;
  return d;
}

int sfmx_fail(sfmx_ctx* ctx, int status, const char* what, hipError_t e);
int sfmx_pyramid_settle(sfmx_ctx* ctx, const sfmx_pyramid* pyr);  // image.hip: order the main stream behind an asynchronous build
extern "C" void sfmx_release_graphs(sfmx_ctx* ctx);  // image.hip: drop the hipGraph executables cached for this context

#define SFMX_HIP(ctx, call)                                                         \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) return sfmx_fail((ctx), SFMX_ERR_HIP, #call, e__);       \
  } while (0)

#define SFMX_REQUIRE(ctx, cond)                                                     \
  do {                                                                              \
    if (!(cond)) return sfmx_fail((ctx), SFMX_ERR_INVALID, #cond, hipSuccess);      \
  } while (0)

static inline void prof_begin(sfmx_ctx* c, int kid) {
  c->prof_open = false;
  if (!c->timing || c->prof_n >= SFMX_PROF_PAIRS) return;
  hipEvent_t* e = c->pev[c->prof_n];
  if (!e[0] && (hipEventCreate(&e[0]) != hipSuccess || hipEventCreate(&e[1]) != hipSuccess)) return;
  if (hipEventRecord(e[0], c->stream) != hipSuccess) return;
  c->prof_kid[c->prof_n] = kid;
  c->prof_open = true;
}
static inline void prof_end(sfmx_ctx* c) {
  if (!c->prof_open) return;
  c->prof_open = false;
  if (hipEventRecord(c->pev[c->prof_n][1], c->stream) == hipSuccess) c->prof_n++;
}
static inline void prof_collect(sfmx_ctx* c) {  // after the stream has been synchronised
  for (int i = 0; i < c->prof_n; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->pev[i][0], c->pev[i][1]) == hipSuccess) {
      c->kus[c->prof_kid[i]] += (double)ms * 1000.0;
      c->kcalls[c->prof_kid[i]]++;
    }
  }
  c->prof_n = 0;
}
// SFMX_PROF(ctx, KID_x, kernel<<<...>>>(...));
#define SFMX_PROF(ctx, kid, launch) \
  do {                              \
    prof_begin((ctx), (kid));       \
    launch;                         \
    prof_end((ctx));                \
  } while (0)

// event timing of the dominant kernel of an API call (only when ctx->timing)
struct KernelTimer {
  sfmx_ctx* c;
  explicit KernelTimer(sfmx_ctx* ctx) : c(ctx) {}
  void start() {
    if (c->timing) (void)hipEventRecord(c->ev0, c->stream);
  }
  void stop() {
    if (c->timing) (void)hipEventRecord(c->ev1, c->stream);
  }
  void collect() {  // call after the stream has been synchronised
    if (c->timing) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_us = (double)ms * 1000.0;
      prof_collect(c);
    }
  }
};
