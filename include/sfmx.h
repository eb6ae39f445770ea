/* sfmx.h — C ABI of the MI355X-native SfM hot path (libsfmx.so, gfx950).
 *
 * The reference (RoozbehSanaei/Structure-from-Motion-3D-Reconstruction, cpp/) has no plugin / FFI
 * layer: its hot functions are file-static in cpp/src/templering_sfm.cpp ("T:" below).  Each entry
 * point here replaces one of those function seams and is what a maintainer would
 * bind from the reference's own code (see INTEGRATION.md for the call-site patch).
 *
 * Conventions
 *  - plain C, no exceptions across the boundary; every call returns an sfmx_status (0 = ok).
 *  - caller owns host buffers; the library owns device memory inside the context.
 *  - one HIP stream per context; a context is thread-compatible, not thread-safe (one per thread).
 *  - all floating point is IEEE binary64, evaluated WITHOUT fused multiply-add in the reference's
 *    operation order, so results are bit-identical to the x86-64 reference build.
 *  - arrays of 2-D points are [n][2] doubles (x,y), row-major 3x3 matrices are 9 doubles.
 */
#ifndef SFMX_H
#define SFMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sfmx_status {
  SFMX_OK = 0,
  SFMX_ERR_INVALID = 1,   /* bad argument (null pointer, size <= 0, level out of range, ...)      */
  SFMX_ERR_HIP = 2,       /* a HIP runtime call failed; see sfmx_last_error()                      */
  SFMX_ERR_NO_DEVICE = 3, /* no gfx950 device / extension unusable: callers must fail, not fall back */
  SFMX_ERR_SINGULAR = 4,  /* sfmx_solve_dense: pivot < 1e-15 — where dense.hpp:67 throws           */
  SFMX_ERR_UNSUPPORTED = 5
} sfmx_status;

typedef struct sfmx_ctx sfmx_ctx;
typedef struct sfmx_pyramid sfmx_pyramid;

/* ---- context ------------------------------------------------------------------------------ */
int sfmx_ctx_create(int device_id, sfmx_ctx** out);
/* same, with a stream priority hint: <0 high (short latency-critical launches), 0 normal, >0 low (background) */
int sfmx_ctx_create_prio(int device_id, int priority, sfmx_ctx** out);
void sfmx_ctx_destroy(sfmx_ctx* ctx);
const char* sfmx_last_error(const sfmx_ctx* ctx);
int sfmx_sync(sfmx_ctx* ctx);
int sfmx_ctx_device(const sfmx_ctx* ctx);  /* device index the context was created on */
/* HIP's current device is per host thread: a thread that did not create the context calls this once before
 * using it (one context per thread; contexts of several threads may share a device). */
int sfmx_ctx_make_current(sfmx_ctx* ctx);
/* raw hipStream_t of the context (for event timing by the caller) */
void* sfmx_stream(sfmx_ctx* ctx);
/* microseconds of GPU time of the most recent hot kernel launched by the last API call, measured
 * with HIP events on the context stream (0 when timing is disabled, see sfmx_set_timing). */
int sfmx_set_timing(sfmx_ctx* ctx, int enabled);
int sfmx_get_timing(const sfmx_ctx* ctx); /* 1 if enabled (so that helper contexts can inherit the setting) */
double sfmx_last_kernel_us(const sfmx_ctx* ctx);
/* Per-kernel profile of this context while timing is enabled: accumulated GPU microseconds and launch counts of each
 * hot kernel (HIP events recorded on the context's stream around every launch).  Enabling timing starts a fresh
 * profile.  Returns the number of kernel ids; at most cap entries are written; reset != 0 clears after reading. */
int sfmx_kernel_profile(sfmx_ctx* ctx, int reset, int cap, double* us_out, uint64_t* calls_out);
const char* sfmx_kernel_profile_name(int id);

/* ---- image pyramid: replaces sfm::GrayImage + build_pyr/downsample2 (T:200-232) ------------ */
/* level 0 is the image itself; level l is (w>>l) x (h>>l), 2x2 box with integer /4 truncation and
 * +1 neighbours clamped to the edge.  All levels live in one HBM allocation. */
int sfmx_pyramid_create(sfmx_ctx* ctx, int w, int h, int levels, sfmx_pyramid** out);
void sfmx_pyramid_destroy(sfmx_ctx* ctx, sfmx_pyramid* pyr);
/* host pixels -> HBM, then build levels 1.. on the device */
int sfmx_pyramid_upload(sfmx_ctx* ctx, sfmx_pyramid* pyr, const uint8_t* host_pixels);
/* pixels already resident in HBM (device pointer): device copy + build */
int sfmx_pyramid_set_device(sfmx_ctx* ctx, sfmx_pyramid* pyr, const void* device_pixels);
/* The same build on the context's SECOND stream (it overlaps with kernels already queued on the first, e.g. the KLT launch of
 * the previous frame); fetch_level >= 0 also copies that level's pixels to pinned host memory.  Nothing that still reads the
 * pyramid may be in flight.  sfmx_pyramid_wait orders the context's main stream behind the build (no host wait; calls of THIS
 * context that take the pyramid do it themselves, another context that reads it must be started after a host-side wait such as
 * sfmx_pyramid_fetched_level or sfmx_sync); sfmx_pyramid_fetched_level waits for the copy and returns the pixels (*out == NULL
 * if that level was not fetched). */
int sfmx_pyramid_set_device_async(sfmx_ctx* ctx, sfmx_pyramid* pyr, const void* device_pixels, int fetch_level);
int sfmx_pyramid_wait(sfmx_ctx* ctx, sfmx_pyramid* pyr);
int sfmx_pyramid_fetched_level(sfmx_ctx* ctx, sfmx_pyramid* pyr, int level, const uint8_t** out);
int sfmx_pyramid_download_level(sfmx_ctx* ctx, const sfmx_pyramid* pyr, int level, uint8_t* host_out);
int sfmx_pyramid_level_size(const sfmx_pyramid* pyr, int level, int* w, int* h);

/* ---- Shi-Tomasi score map: replaces the per-pixel loop of shi_tomasi (T:242-272) ----------- */
/* score_out [h][w] doubles (0 outside the r=2 interior band); max_out = max over the map (T:274).
 * Thresholding, std::sort and the greedy min-distance pick (T:275-301) stay on the host because
 * their result depends on libstdc++'s sort permutation. */
int sfmx_shi_tomasi_score(sfmx_ctx* ctx, const sfmx_pyramid* pyr, double* score_out, double* max_out);
/* candidates only: pixels with score >= max*quality in row-major order (T:280-285), as
 * (x | y<<16) in cand_xy and the score in cand_score.  *n_out is the total number of candidates;
 * at most cap are written. */
int sfmx_shi_tomasi_candidates(sfmx_ctx* ctx, const sfmx_pyramid* pyr, double quality, int cap,
                               uint32_t* cand_xy, double* cand_score, int* n_out, double* max_out);

/* Same, after resolving on the device every candidate whose fate under the greedy min-distance pick
 * (T:288-300) is certain whatever the sort's tie order (parallel fixpoint: "accepted" once every pixel
 * within min_dist with score >= its own is rejected; "rejected" once an accepted pixel of strictly
 * greater score lies within min_dist).  Rejected candidates are dropped; bit 31 of cand_xy marks the
 * certainly accepted ones, the rest are still undecided; x = bits 0..14, y = bits 16..30.
 * *n_out = survivors (row-major order), *n_total_out = all candidates before resolution. */
int sfmx_shi_tomasi_candidates_pruned(sfmx_ctx* ctx, const sfmx_pyramid* pyr, double quality, int min_dist,
                                      int cap, uint32_t* cand_xy, double* cand_score, int32_t* cand_full_index,
                                      int* n_out, int* n_total_out, double* max_out);
/* cand_full_index[k] (optional) = position of survivor k in the row-major list of ALL candidates.  That list
 * is kept as 16-byte records {double score; uint32 id (= position); uint32 mark (= 0)} and is downloaded
 * speculatively into pinned host memory owned by the context; *keys_out stays valid (and may be modified
 * in place) until the next Shi-Tomasi call on this context. */
int sfmx_shi_tomasi_fetch_all_keys(sfmx_ctx* ctx, int n_total, void** keys_out);

/* ---- KLT: replaces KLTTracker::track_one fwd+bwd and the FB test (T:356-362, 402-460) ------- */
typedef struct sfmx_klt_cfg {
  int levels;       /* LKConfig::pyr_levels (T:312) */
  int win_radius;   /* LKConfig::win_radius (T:313); supported 1..7 */
  int iters;        /* LKConfig::iters (T:314) */
  double fb_thresh; /* LKConfig::fb_thresh (T:315) */
} sfmx_klt_cfg;
/* For every point: fwd = track_one(A,B,p), back = track_one(B,A,fwd), keep = !(hypot(back-p) >= fb).
 * xy_back may be NULL.  n_steps_out (optional) receives the number of lk_step evaluations executed. */
int sfmx_klt_track(sfmx_ctx* ctx, const sfmx_pyramid* pyr_a, const sfmx_pyramid* pyr_b,
                   const double* xy_in, int n, const sfmx_klt_cfg* cfg, double* xy_fwd,
                   double* xy_back, uint8_t* keep, uint64_t* n_steps_out);

/* ---- RANSAC scoring: replaces the hypothesis loop of find_E_ransac (T:664-677) --------------- */
/* xi,xj: K^-1-normalised correspondences [n][2]; idx8: pre-drawn sample octets [H][8] (the
 * caller draws them with the libstdc++-compatible generator so the stream matches T:657-665).
 * Every 8-point hypothesis (T:609-627) is built and all n points are scored against it with the
 * Sampson error (T:629-638, bit-exact arithmetic for a given E), counting err < thr.
 *
 * Which E a hypothesis is scored with.  The reference's eight_point_E calls the platform libm
 * (atan2/cos/sin, linalg.hpp:156-157); the device runs the same Jacobi with algebraic rotations and
 * reports a conditioning estimate cond per hypothesis (relative gap of the two smallest
 * eigenvalues, relative singular-value gap of the rank-2 projection; 0 if a Jacobi pivot was
 * nearly tied with another entry).  Its E differs from the reference's by at most 1e-16 / cond per
 * entry (measured <= 3e-18 / cond).  Octets with a repeated sample index (sampling is with
 * replacement, T:665: null space of dimension >= 2, cond ~ 0) and hypotheses with cond < 1e-13
 * are derived on the host with libm instead (the reference's E bit for bit, flags bit 0) and
 * scored with that.
 *
 * counts_out [H]: #{err < thr}.  lo_out/hi_out [H] (optional): bounds of the REFERENCE's count of
 * that iteration under the EMPIRICAL contract above (|E_dev - E_ref| <= 1e-16 / cond: measured, not
 * derived; a caller that needs the reference's winner verifies candidates exactly, as
 * ransac_local in csrc/host/pipeline.cpp does, and falls back to exact counts if a bound fails): a point counts for lo only if it stays an inlier, for hi
 * if it can become one, when every entry of E moves by 1e-16 / cond (per-point bound, see
 * k_score); lo == hi == count for the exact hypotheses.  flags_out [H] (optional): bit 0 = exact
 * host hypothesis.  cond_out [H] (optional): the conditioning estimate (+inf for exact ones).
 * best_iter/best_count = argmax of counts with the LOWEST iteration on ties (the reference's
 * strict '>').  E_out [H][9] (optional): the hypotheses that were scored. */
int sfmx_ransac_score_ex(sfmx_ctx* ctx, const double* xi, const double* xj, int n, const int32_t* idx8,
                         int H, double thr, int32_t* counts_out, int32_t* lo_out, int32_t* hi_out,
                         uint8_t* flags_out, double* cond_out, int32_t* best_iter,
                         int32_t* best_count, double* E_out);
/* same without the certification outputs */
int sfmx_ransac_score(sfmx_ctx* ctx, const double* xi, const double* xj, int n, const int32_t* idx8,
                      int H, double thr, int32_t* counts_out, int32_t* best_iter, int32_t* best_count,
                      double* E_out);
/* inlier mask of ONE essential matrix (used for the winner after the host has re-derived its E
 * with the platform libm, so the mask and E are bit-identical to the reference's).  Passing
 * xi == xj == NULL reuses the n correspondences left in HBM by the preceding sfmx_ransac_score. */
int sfmx_sampson_mask(sfmx_ctx* ctx, const double* xi, const double* xj, int n, const double* E9,
                      double thr, uint8_t* mask_out, int32_t* count_out);

/* ---- local BA: replaces the S,b build of bundle_adjust_window (T:893-1071) ------------------ */
/* poses_wc [W][12] = world->camera (R row-major, t); X [P][3]; CSR observations: obs_ptr [P+1],
 * obs_li [R] window-local pose index, obs_uv [R][2] pixels.  Points are consumed in array order
 * (the caller passes them in the reference's unordered_map iteration order).
 * Output: S [6W][6W] row-major and b [6W] AFTER damping (+lambda on the diagonal) and gauge
 * (+1e9 on DoF 0..5, b[0..5] = 0) when damp != 0.  Accumulation order is the reference's, so S,b
 * are bit-identical on one GPU. */
typedef struct sfmx_ba_problem sfmx_ba_problem;
int sfmx_ba_create(sfmx_ctx* ctx, int W, int P, const double* X, const int32_t* obs_ptr,
                   const int32_t* obs_li, const double* obs_uv, sfmx_ba_problem** out);
/* re-target an existing problem object at new data (device buffers are kept and only grow) */
int sfmx_ba_reset(sfmx_ctx* ctx, sfmx_ba_problem* prob, int W, int P, const double* X, const int32_t* obs_ptr,
                  const int32_t* obs_li, const double* obs_uv);
void sfmx_ba_destroy(sfmx_ctx* ctx, sfmx_ba_problem* prob);
int sfmx_ba_build(sfmx_ctx* ctx, sfmx_ba_problem* prob, const double* poses_wc, double fx, double fy,
                  double cx, double cy, double huber, double lambda, int damp, double* S_out,
                  double* b_out);
/* build + solve on the device in one submission: dx [6W]; returns SFMX_ERR_SINGULAR where the
 * reference's solve_gauss would throw (the caller then skips BA as T:1076-1078 does). */
int sfmx_ba_step(sfmx_ctx* ctx, sfmx_ba_problem* prob, const double* poses_wc, double fx, double fy,
                 double cx, double cy, double huber, double lambda, double* dx_out);
/* Optional bracket around the sfmx_ba_step calls of ONE bundle_adjust_window call (T:893-1096: `iters` iterations on the same
 * window and intrinsics).  For window-sized problems sfmx_ba_begin launches a kernel that stays resident for the whole job, so
 * that an iteration is a hand-over of poses and a poll for S | b instead of two launches that queue behind the kernels of other
 * contexts; sfmx_ba_end releases it when the job stops early (SFMX_ERR_SINGULAR).  Results are identical with and without. */
int sfmx_ba_begin(sfmx_ctx* ctx, sfmx_ba_problem* prob, int iters, double fx, double fy, double cx, double cy, double huber,
                  double lambda);
int sfmx_ba_end(sfmx_ctx* ctx, sfmx_ba_problem* prob);
/* partial sums for point-sharded multi-GPU BA: raw S,b of this problem's points only (no damping);
 * device pointers (valid until the next call on prob) so the caller can all-reduce them in HBM. */
int sfmx_ba_build_partial(sfmx_ctx* ctx, sfmx_ba_problem* prob, const double* poses_wc, double fx,
                          double fy, double cx, double cy, double huber, void** S_dev, void** b_dev);

/* ---- multi-GPU exchange steps (one process per GPU, RCCL over xGMI) ---------------------------- */
/* The reference is single-process; these are the collectives the sharded modes of this library add.
 * A communicator belongs to ONE host thread / context at a time (a pipeline uses one per lane).  world == 1 (or a NULL
 * communicator) makes every call below a local no-op, so the sharded entry points can be used unconditionally. */
typedef struct sfmx_comm sfmx_comm;
#define SFMX_COMM_ID_BYTES 128
int sfmx_comm_get_unique_id(void* id_out);  /* rank 0; the application carries the 128 bytes to the other ranks */
int sfmx_comm_create(int device_id, const void* id_bytes, int rank, int world, sfmx_comm** out);
void sfmx_comm_destroy(sfmx_comm* comm);
int sfmx_comm_rank(const sfmx_comm* comm);
int sfmx_comm_world(const sfmx_comm* comm);
/* contiguous, order-preserving split of range(n): the first n % world ranks get one item more */
void sfmx_shard_range(int n, int rank, int world, int* lo, int* hi);
/* all-reduce of a small host array through the context's stream; op 0 = sum, 1 = max */
int sfmx_comm_allreduce_f64(sfmx_ctx* ctx, sfmx_comm* comm, double* host_inout, int n, int op);
int sfmx_comm_allreduce_u64_max(sfmx_ctx* ctx, sfmx_comm* comm, uint64_t* host_inout, int n);
/* Point-sharded BA iteration (T:893-1095): prob holds THIS rank's contiguous range of the window's points (reference
 * order); raw S | b of the shard -> one all-reduce(sum) of D*D + D doubles in HBM -> damping + gauge (T:1064-1071) ->
 * solve on the device -> dx (identical on every rank).  TOLERANCE mode: the rank-ordered sum rounds differently from the
 * sequential reference (1e-9 relative per step; with world == 1 it equals sfmx_ba_step bit for bit), and whole runs of the
 * reference's BA do NOT stay within the task's 1e-6 of the one-GPU run (tools/virtual_world_probe.py).  The pipeline uses
 * sfmx_ba_step_sharded_elements below; this entry remains for callers that accept the tolerance (SFMX_BA_SHARD=points). */
int sfmx_ba_step_sharded(sfmx_ctx* ctx, sfmx_comm* comm, sfmx_ba_problem* prob, const double* poses_wc, double fx,
                         double fy, double cx, double cy, double huber, double lambda, double* dx_out);

/* Element-sharded BA iteration (T:893-1095) -- the sharded mode that keeps the reference's arithmetic.  prob holds the WHOLE
 * window on every rank.  The per-point records are computed by every rank (replicated); rank r forms and reduces only its
 * contiguous slice of the elements of S | b (every element's sum runs over all points in the reference's order) and contributes
 * +0.0 elsewhere, so the all-reduce(sum) of D*D + D doubles only ever adds zeros: S, b and dx are bit-identical to sfmx_ba_step
 * at any world size.  (sfmx_ba_step_sharded regroups the addends instead; the reference's BA amplifies that rounding
 * difference -- it ADDS the Schur term, T:1055 -- to a different trajectory within tens of keyframes) */
int sfmx_ba_step_sharded_elements(sfmx_ctx* ctx, sfmx_comm* comm, sfmx_ba_problem* prob, const double* poses_wc, double fx,
                                  double fy, double cx, double cy, double huber, double lambda, double* dx_out);

/* ---- dense solve: replaces sfm::solve_gauss (cpp/include/dense.hpp:54-93) -------------------- */
/* Gaussian elimination with partial pivoting in the reference's operation order; A [n][n]
 * row-major and b [n] are not modified; x [n].  SFMX_ERR_SINGULAR when a pivot is < 1e-15. */
int sfmx_solve_dense(sfmx_ctx* ctx, const double* A, const double* b, int n, double* x);

/* lk_step evaluations of the last sfmx_klt_track call that could not use the shared sample grid (more distinct
 * sample coordinates than its descriptor table holds) and took the per-pixel path: a performance counter, the
 * results are identical either way. */
uint64_t sfmx_debug_klt_slow_steps(const sfmx_ctx* ctx);

/* ---- pose-graph normal equations, structured (TOLERANCE mode): replaces the dense 3N x 3N solve of
 * posegraph_optimize_centers (T:1131-1197 -> dense.hpp:54-93) for large keyframe counts ------------ */
/* The system is H = L (x) I_3 with L the N x N weighted graph Laplacian plus the gauge term, so the three coordinates
 * are solved together on L: blocked Cholesky with the trailing update on the FP64 matrix cores
 * (v_mfma_f64_16x16x4_f64), then blocked triangular solves.  entry_ij [m][2] / entry_v [m]: the distinct entries of
 * the LOWER triangle of L (row >= column), already summed; g3, x3 [n][3].  A different factorisation than the
 * reference's elimination: agrees with solve_gauss on the dense system to ~1e-12 relative (tests: 1e-9), not bit for
 * bit.  SFMX_ERR_SINGULAR where a pivot is not > 1e-15 (a keyframe not connected to node 0: the reference throws). */
int sfmx_posegraph_solve(sfmx_ctx* ctx, int n, const int32_t* entry_ij, const double* entry_v, int m,
                         const double* g3, double* x3);

/* ---- self-check hooks used by the parity tests (device arithmetic vs the host libm) ---------- */
int sfmx_debug_hypot(sfmx_ctx* ctx, const double* x, const double* y, int n, double* out);
int sfmx_debug_divsqrt(sfmx_ctx* ctx, const double* x, const double* y, int n, double* div_out,
                       double* sqrt_out);

#ifdef __cplusplus
}
#endif
#endif /* SFMX_H */
