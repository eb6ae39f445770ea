// posegraph.hip — structured solve of the pose-graph normal equations for large keyframe counts (gfx950, FP64 MFMA).
//
// posegraph_optimize_centers (reference cpp/src/templering_sfm.cpp T:1131-1197) builds the dense 3N x 3N system
//   H = sum_edges w (e_i - e_j)(e_i - e_j)^T (x) I_3  +  1e9 on the three DoF of node 0,      H dc = g
// and hands it to sfm::solve_gauss (dense.hpp:54-93): O((3N)^3) time and (3N)^2 doubles -- 7.2 GB and hours at the
// 10k keyframes of BASELINE config 5.  The three coordinates never mix: H = L (x) I_3 with L the N x N weighted graph
// Laplacian (+ gauge), symmetric positive definite whenever every keyframe is connected to node 0.  This file solves
// L X = G for the three right-hand sides at once:
//
//   blocked right-looking Cholesky, 32 columns per step:
//     k_chol_panel   every workgroup factors the 32 x 32 diagonal block in LDS (redundantly: no inter-workgroup hand-off),
//                    then solves its rows of the panel against it, one row per thread;
//     k_chol_update  trailing update A22 -= L21 L21^T on the lower triangle with v_mfma_f64_16x16x4_f64 (64 x 64 tile
//                    per workgroup, one 32 x 32 quadrant = 2 x 2 MFMA tiles x 8 k-steps per wavefront, operands in LDS);
//   k_tri_forward / k_tri_backward: blocked triangular solves with the three right-hand sides.
//
// This is the library's TOLERANCE mode for this system: a different factorisation order than the reference's Gaussian
// elimination, fused multiply-adds in the matrix cores.  It agrees with solve_gauss on the dense system to ~1e-12
// relative (tests hold it to 1e-9; graph Laplacians are far better conditioned in practice than their condition number
// suggests), which is orders of magnitude inside the ATE tolerance of 1e-6.  The bit-exact dense path (ba.hip) stays the
// default for the sizes the reference can run.  Singular systems (a keyframe not connected to node 0: the reference's
// elimination meets a pivot < 1e-15 and throws) are reported as SFMX_ERR_SINGULAR.
#include "sfmx_internal.h"

#define PG_NB 32   // columns per factorisation step
#define PG_TILE 64 // trailing-update tile per workgroup

typedef double pg_f64x4 __attribute__((ext_vector_type(4)));

// A = sum over the (host-merged, unique) lower-triangle entries
__global__ void k_pg_scatter(double* __restrict__ A, int ld, const int32_t* __restrict__ ij, const double* __restrict__ v, int m) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  A[(size_t)ij[2 * e] * ld + ij[2 * e + 1]] = v[e];
}

// unblocked Cholesky of the diagonal block held in LDS (lower triangle; D[r][c], r >= c); all threads of the workgroup
// take part.  status |= 1 when a pivot is not > 1e-15 (singular / not positive definite).
__device__ void chol_block_lds(double (*D)[PG_NB + 1], int nb, int* bad) {
  for (int j = 0; j < nb; j++) {
    __syncthreads();
    const double djj = D[j][j];
    if (!(djj > 1e-15)) {
      if (threadIdx.x == 0) *bad = 1;
      __syncthreads();
      return;
    }
    const double inv = 1.0 / sqrt(djj);
    __syncthreads();
    for (int r = j + threadIdx.x; r < nb; r += blockDim.x) D[r][j] = (r == j) ? sqrt(djj) : D[r][j] * inv;
    __syncthreads();
    // trailing update of the block: D[r][c] -= D[r][j] * D[c][j] for j < c <= r
    const int rem = nb - j - 1;
    for (int e = threadIdx.x; e < rem * rem; e += blockDim.x) {
      const int r = j + 1 + e / rem, c = j + 1 + e % rem;
      if (c <= r) D[r][c] -= D[r][j] * D[c][j];
    }
  }
  __syncthreads();
}

// step k0: factor A[k0:k1, k0:k1] -> Ldiag (workgroup 0 stores it), rows below: L21 = A21 L11^-T (in place)
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int n, int ld, int k0, double* __restrict__ Ldiag, int* __restrict__ status) {
  __shared__ double D[PG_NB][PG_NB + 1];
  __shared__ int bad;
  if (status[0]) return;
  const int nb = min(PG_NB, n - k0);
  if (threadIdx.x == 0) bad = 0;
  for (int e = threadIdx.x; e < PG_NB * PG_NB; e += blockDim.x) {
    const int r = e / PG_NB, c = e % PG_NB;
    D[r][c] = (r < nb && c <= r) ? A[(size_t)(k0 + r) * ld + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  chol_block_lds(D, nb, &bad);
  if (bad) {
    if (blockIdx.x == 0 && threadIdx.x == 0) status[0] = 1;
    return;
  }
  if (blockIdx.x == 0)
    for (int e = threadIdx.x; e < PG_NB * PG_NB; e += blockDim.x) Ldiag[e] = D[e / PG_NB][e % PG_NB];
  const int r = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double* row = A + (size_t)r * ld + k0;
  double x[PG_NB];
#pragma unroll
  for (int c = 0; c < PG_NB; c++) x[c] = c < nb ? row[c] : 0.0;
#pragma unroll
  for (int c = 0; c < PG_NB; c++) {
    if (c < nb) {
      double s = x[c];
#pragma unroll
      for (int t = 0; t < PG_NB; t++)
        if (t < c) s -= x[t] * D[c][t];
      x[c] = s / D[c][c];
    }
  }
#pragma unroll
  for (int c = 0; c < PG_NB; c++)
    if (c < nb) row[c] = x[c];
}

// A22 -= L21 L21^T on the lower triangle, 64 x 64 tile per workgroup (4 wavefronts, 32 x 32 each), K = 32.
// v_mfma_f64_16x16x4_f64: A[l & 15][k = l >> 4], B[k = l >> 4][l & 15], C/D col = l & 15, row = (l >> 4) + 4 reg.
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ A, int n, int ld, int k0, const int* __restrict__ status) {
  __shared__ double Li[PG_TILE][PG_NB + 1];  // rows of the output tile
  __shared__ double Lj[PG_TILE][PG_NB + 1];  // columns of the output tile (rows of L21 as well: L21^T)
  if (status[0]) return;
  const int k1 = k0 + PG_NB;  // only called while k1 < n (full panel)
  // lower-triangle tile pair from the linear block index
  int bi = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= (int)blockIdx.x) bi++;
  while (bi * (bi + 1) / 2 > (int)blockIdx.x) bi--;
  const int bj = blockIdx.x - bi * (bi + 1) / 2;
  const int i0 = k1 + bi * PG_TILE, j0 = k1 + bj * PG_TILE;
  for (int e = threadIdx.x; e < PG_TILE * PG_NB; e += blockDim.x) {
    const int r = e / PG_NB, c = e % PG_NB;
    Li[r][c] = (i0 + r < n) ? A[(size_t)(i0 + r) * ld + k0 + c] : 0.0;
    Lj[r][c] = (j0 + r < n) ? A[(size_t)(j0 + r) * ld + k0 + c] : 0.0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;  // this wavefront's 32 x 32 quadrant
  if (bi == bj && wj > wi) return;                        // strictly upper quadrant of a diagonal tile
  pg_f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = (pg_f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < PG_NB / 4; kk++) {
    double av[2], bv[2];
#pragma unroll
    for (int a = 0; a < 2; a++) av[a] = Li[wi + 16 * a + (lane & 15)][4 * kk + (lane >> 4)];
#pragma unroll
    for (int b = 0; b < 2; b++) bv[b] = Lj[wj + 16 * b + (lane & 15)][4 * kk + (lane >> 4)];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int reg = 0; reg < 4; reg++) {
        const int r = i0 + wi + 16 * a + (lane >> 4) + 4 * reg, c = j0 + wj + 16 * b + (lane & 15);
        if (r < n && c <= r) A[(size_t)r * ld + c] -= acc[a][b][reg];
      }
}

// forward substitution step: y_k = L11^-1 g_k (every workgroup, redundantly; workgroup 0 stores it to Y), then
// g_r -= L21[r] . y_k for its rows r below the block.  G, Y: [n][4] (3 right-hand sides, padded)
__global__ __launch_bounds__(256) void k_tri_forward(const double* __restrict__ A, int n, int ld, int k0, const double* __restrict__ Ldiag,
                                                     double* __restrict__ G, double* __restrict__ Y) {
  __shared__ double D[PG_NB][PG_NB + 1];
  __shared__ double y[PG_NB][3];
  const int nb = min(PG_NB, n - k0);
  for (int e = threadIdx.x; e < PG_NB * PG_NB; e += blockDim.x) D[e / PG_NB][e % PG_NB] = Ldiag[e];
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    for (int c = 0; c < nb; c++) {
      double s = G[(size_t)(k0 + c) * 4 + d];
      for (int t = 0; t < c; t++) s -= D[c][t] * y[t][d];
      y[c][d] = s / D[c][c];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < nb)
    for (int d = 0; d < 3; d++) Y[(size_t)(k0 + threadIdx.x) * 4 + d] = y[threadIdx.x][d];
  const int r = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double* row = A + (size_t)r * ld + k0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < nb; c++) {
    const double l = row[c];
    s0 += l * y[c][0]; s1 += l * y[c][1]; s2 += l * y[c][2];
  }
  G[(size_t)r * 4 + 0] -= s0; G[(size_t)r * 4 + 1] -= s1; G[(size_t)r * 4 + 2] -= s2;
}

// backward substitution step (L^T x = y), blocks from the last to the first: x_k = L11^-T y_k (redundantly; workgroup 0
// stores it to X), then y_c -= sum_{r in block} L[r][c] x_r for its columns c < k0
__global__ __launch_bounds__(256) void k_tri_backward(const double* __restrict__ A, int n, int ld, int k0, const double* __restrict__ Ldiag,
                                                      double* __restrict__ Y, double* __restrict__ X) {
  __shared__ double D[PG_NB][PG_NB + 1];
  __shared__ double x[PG_NB][3];
  const int nb = min(PG_NB, n - k0);
  for (int e = threadIdx.x; e < PG_NB * PG_NB; e += blockDim.x) D[e / PG_NB][e % PG_NB] = Ldiag[e];
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    for (int c = nb - 1; c >= 0; c--) {
      double s = Y[(size_t)(k0 + c) * 4 + d];
      for (int t = c + 1; t < nb; t++) s -= D[t][c] * x[t][d];
      x[c][d] = s / D[c][c];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < nb)
    for (int d = 0; d < 3; d++) X[(size_t)(k0 + threadIdx.x) * 4 + d] = x[threadIdx.x][d];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k0) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < nb; r++) {
    const double l = A[(size_t)(k0 + r) * ld + c];
    s0 += l * x[r][0]; s1 += l * x[r][1]; s2 += l * x[r][2];
  }
  Y[(size_t)c * 4 + 0] -= s0; Y[(size_t)c * 4 + 1] -= s1; Y[(size_t)c * 4 + 2] -= s2;
}

extern "C" int sfmx_posegraph_solve(sfmx_ctx* c, int n, const int32_t* entry_ij, const double* entry_v, int m, const double* g3, double* x3) {
  SFMX_REQUIRE(c, c && n >= 1 && entry_ij && entry_v && m >= 1 && g3 && x3);
  for (int e = 0; e < m; e++) SFMX_REQUIRE(c, entry_ij[2 * e] >= 0 && entry_ij[2 * e] < n && entry_ij[2 * e + 1] >= 0 && entry_ij[2 * e + 1] <= entry_ij[2 * e]);
  const int ld = (n + 63) & ~63;
  const int steps = (n + PG_NB - 1) / PG_NB;
  const size_t abytes = (size_t)n * ld * 8, dbytes = (size_t)steps * PG_NB * PG_NB * 8, vbytes = (size_t)n * 32;
  c->resident_points = 0;
  SFMX_HIP(c, c->d[7].ensure(abytes + dbytes + 3 * vbytes + 64));
  SFMX_HIP(c, c->d[0].ensure((size_t)m * 16));
  SFMX_HIP(c, c->h[0].ensure((size_t)m * 16 + vbytes));
  double* A = c->d[7].as<double>();
  double* Ld = reinterpret_cast<double*>(c->d[7].as<char>() + abytes);
  double* G = reinterpret_cast<double*>(c->d[7].as<char>() + abytes + dbytes);
  double* Y = G + (size_t)n * 4;
  double* X = Y + (size_t)n * 4;
  int* status = reinterpret_cast<int*>(X + (size_t)n * 4);
  char* hs = c->h[0].as<char>();
  memcpy(hs, entry_ij, (size_t)m * 8);
  memcpy(hs + (size_t)m * 8, entry_v, (size_t)m * 8);
  double* hg = reinterpret_cast<double*>(hs + (size_t)m * 16);
  for (int i = 0; i < n; i++) { hg[4 * i] = g3[3 * i]; hg[4 * i + 1] = g3[3 * i + 1]; hg[4 * i + 2] = g3[3 * i + 2]; hg[4 * i + 3] = 0.0; }
  SFMX_HIP(c, hipMemsetAsync(A, 0, abytes, c->stream));
  SFMX_HIP(c, hipMemsetAsync(status, 0, 64, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(c->d[0].p, hs, (size_t)m * 16, hipMemcpyHostToDevice, c->stream));
  SFMX_HIP(c, hipMemcpyAsync(G, hg, vbytes, hipMemcpyHostToDevice, c->stream));
  KernelTimer t(c);
  t.start();
  prof_begin(c, KID_SOLVE);
  k_pg_scatter<<<(m + 255) / 256, 256, 0, c->stream>>>(A, ld, c->d[0].as<int32_t>(), reinterpret_cast<const double*>(c->d[0].as<char>() + (size_t)m * 8), m);
  for (int s = 0; s < steps; s++) {
    const int k0 = s * PG_NB, nb = (n - k0) < PG_NB ? (n - k0) : PG_NB, below = n - k0 - nb;
    k_chol_panel<<<below > 0 ? (below + 255) / 256 : 1, 256, 0, c->stream>>>(A, n, ld, k0, Ld + (size_t)s * PG_NB * PG_NB, status);
    if (below > 0) {
      const int tiles = (below + PG_TILE - 1) / PG_TILE;
      k_chol_update<<<tiles * (tiles + 1) / 2, 256, 0, c->stream>>>(A, n, ld, k0, status);
    }
  }
  for (int s = 0; s < steps; s++) {
    const int k0 = s * PG_NB, nb = (n - k0) < PG_NB ? (n - k0) : PG_NB, below = n - k0 - nb;
    k_tri_forward<<<below > 0 ? (below + 255) / 256 : 1, 256, 0, c->stream>>>(A, n, ld, k0, Ld + (size_t)s * PG_NB * PG_NB, G, Y);
  }
  for (int s = steps - 1; s >= 0; s--) {
    const int k0 = s * PG_NB;
    k_tri_backward<<<k0 > 0 ? (k0 + 255) / 256 : 1, 256, 0, c->stream>>>(A, n, ld, k0, Ld + (size_t)s * PG_NB * PG_NB, Y, X);
  }
  prof_end(c);
  t.stop();
  SFMX_HIP(c, hipGetLastError());
  SFMX_HIP(c, c->h[1].ensure(vbytes + 64));
  SFMX_HIP(c, hipMemcpyAsync(c->h[1].p, X, vbytes + 64, hipMemcpyDeviceToHost, c->stream));
  SFMX_HIP(c, hipStreamSynchronize(c->stream));
  t.collect();
  const double* hx = c->h[1].as<double>();
  int st = 0;
  memcpy(&st, c->h[1].as<char>() + vbytes, 4);
  if (st) return SFMX_ERR_SINGULAR;
  for (int i = 0; i < n; i++) { x3[3 * i] = hx[4 * i]; x3[3 * i + 1] = hx[4 * i + 1]; x3[3 * i + 2] = hx[4 * i + 2]; }
  return SFMX_OK;
}
