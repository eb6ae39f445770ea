// ctx.hip — context lifetime, error strings, debug hooks.
#include "sfmx_internal.h"

int sfmx_fail(sfmx_ctx* ctx, int status, const char* what, hipError_t e) {
  if (ctx) {
    ctx->err = what ? what : "";
    if (e != hipSuccess) {
      ctx->err += ": ";
      ctx->err += hipGetErrorString(e);
    }
  }
  return status;
}

extern "C" {

int sfmx_ctx_create(int device_id, sfmx_ctx** out) { return sfmx_ctx_create_prio(device_id, 0, out); }

int sfmx_ctx_create_prio(int device_id, int priority, sfmx_ctx** out) {
  if (!out) return SFMX_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return SFMX_ERR_NO_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return SFMX_ERR_NO_DEVICE;
  sfmx_ctx* c = new sfmx_ctx;
  c->device = device_id;
  // priority: <0 latency-critical chains of tiny kernels (BA), >0 background work with large grids (corner prefetch)
  int lo = 0, hi = 0;  // HIP: numerically lower = higher priority; range [hi, lo]
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  const int prio = priority < 0 ? hi : (priority > 0 ? lo : (lo + hi) / 2);
  // HIP hands streams to its hardware queues (GPU_MAX_HW_QUEUES, set to 8 by the CLI and bench.py) in creation order.
  // With the copy stream created FIRST, the compute streams of the five pipeline contexts land on queues that do not
  // share with each other (only lane C, which the geometry lane waits for anyway, doubles up with it); the opposite order
  // puts the BA lane next to the prefetch lane and costs 15-20 % keyframes/s (measured on MI355X).
  if (hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio) != hipSuccess ||
      hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    delete c;
    return SFMX_ERR_HIP;
  }
  *out = c;
  return SFMX_OK;
}

void sfmx_ctx_destroy(sfmx_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  sfmx_release_graphs(c);
  if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
  for (auto& b : c->d) b.release();
  for (auto& b : c->wl) b.release();
  for (auto& b : c->h) b.release();
  for (auto& e : c->pev)
    for (auto& x : e)
      if (x) (void)hipEventDestroy(x);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  delete c;
}

const char* sfmx_last_error(const sfmx_ctx* c) { return c ? c->err.c_str() : "null context"; }

int sfmx_sync(sfmx_ctx* c) {
  if (!c) return SFMX_ERR_INVALID;
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  return SFMX_OK;
}
int sfmx_ctx_device(const sfmx_ctx* c) { return c ? c->device : -1; }
int sfmx_ctx_make_current(sfmx_ctx* c) {
  if (!c) return SFMX_ERR_INVALID;
  SFMX_HIP(c, hipSetDevice(c->device));
  return SFMX_OK;
}
void* sfmx_stream(sfmx_ctx* c) { return c ? (void*)c->stream : nullptr; }
int sfmx_set_timing(sfmx_ctx* c, int enabled) {
  if (!c) return SFMX_ERR_INVALID;
  if (enabled && !c->timing) {  // a fresh profile for every timed run
    for (auto& v : c->kus) v = 0.0;
    for (auto& v : c->kcalls) v = 0;
  }
  c->timing = enabled != 0;
  c->prof_n = 0;
  return SFMX_OK;
}
int sfmx_kernel_profile(sfmx_ctx* c, int reset, int cap, double* us_out, uint64_t* calls_out) {
  if (!c) return 0;
  for (int i = 0; i < KID_COUNT && i < cap; i++) {
    if (us_out) us_out[i] = c->kus[i];
    if (calls_out) calls_out[i] = c->kcalls[i];
    if (reset) { c->kus[i] = 0.0; c->kcalls[i] = 0; }
  }
  return KID_COUNT;
}
const char* sfmx_kernel_profile_name(int id) {
  static const char* names[KID_COUNT] = {"k_klt_track", "k_hypotheses", "k_score", "k_ba_points", "k_ba_expand", "k_ba_reduce",
                                         "solve (k_solve_regs / k_solve_wave / k_lu_*)", "k_shi_score",
                                         "shi fixpoint (k_shi_round / k_shi_list_* / k_shi_tail)", "k_downsample2"};
  return (id >= 0 && id < KID_COUNT) ? names[id] : "";
}
int sfmx_get_timing(const sfmx_ctx* c) { return (c && c->timing) ? 1 : 0; }
double sfmx_last_kernel_us(const sfmx_ctx* c) { return c ? c->last_us : 0.0; }

}  // extern "C"

// ---- device arithmetic self-checks -------------------------------------------------------------
__global__ void k_debug_hypot(const double* x, const double* y, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = sfmx::hypot_glibc(x[i], y[i]);
}
__global__ void k_debug_divsqrt(const double* x, const double* y, int n, double* d, double* s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    d[i] = x[i] / y[i];
    s[i] = sqrt(fabs(x[i]));
  }
}

extern "C" {

int sfmx_debug_hypot(sfmx_ctx* c, const double* x, const double* y, int n, double* out) {
  SFMX_REQUIRE(c, c && x && y && out && n > 0);
  const size_t nb = (size_t)n * 8;
  c->resident_points = 0;
  SFMX_HIP(c, c->d[0].ensure(nb));
  SFMX_HIP(c, c->d[1].ensure(nb));
  SFMX_HIP(c, c->d[2].ensure(nb));
  SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, x, nb, hipMemcpyHostToDevice, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(c->d[1].p, y, nb, hipMemcpyHostToDevice, c->stream));
  k_debug_hypot<<<(n + 255) / 256, 256, 0, c->stream>>>(c->d[0].as<double>(), c->d[1].as<double>(), n, c->d[2].as<double>());
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipMemcpyAsync(out, c->d[2].p, nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  return SFMX_OK;
}

int sfmx_debug_divsqrt(sfmx_ctx* c, const double* x, const double* y, int n, double* dv, double* sq) {
  SFMX_REQUIRE(c, c && x && y && dv && sq && n > 0);
  const size_t nb = (size_t)n * 8;
  for (int i = 0; i < 4; i++) SFMX_HIP(c, c->d[i].ensure(nb));
  SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, x, nb, hipMemcpyHostToDevice, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(c->d[1].p, y, nb, hipMemcpyHostToDevice, c->stream));
  k_debug_divsqrt<<<(n + 255) / 256, 256, 0, c->stream>>>(c->d[0].as<double>(), c->d[1].as<double>(), n,
                                                        c->d[2].as<double>(), c->d[3].as<double>());
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, hipMemcpyAsync(dv, c->d[2].p, nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(sq, c->d[3].p, nb, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  return SFMX_OK;
}

}  // extern "C"
