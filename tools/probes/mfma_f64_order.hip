// Does the FP64 matrix core add the k terms of a dot product one after the other, rounding to FP64 after each (i.e. D = fma(a3,b3,
// fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))), and which lane holds which element?  With B = 1.0 such an instruction performs four
// ORDERED additions -- what the ordered sums of an lk_step need (klt.hip).  Build: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

__global__ void k_4x4x4(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
__global__ void k_16x16x4(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 cc = {c[l], c[64 + l], c[128 + l], c[192 + l]};
  d4 r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], cc, 0, 0, 0);
  for (int k = 0; k < 4; k++) d[64 * k + l] = r[k];
}
// chain of dependent instructions: n4 groups of four addends per accumulator, as the kernel would issue them
__global__ void k_chain(const double* a, int n4, double* d, unsigned long long* ticks) {
  const int l = threadIdx.x;
  double acc = 0.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < n4; s++) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[64 * s + l], 1.0, acc, 0, 0, 0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  d[l] = acc;
  if (l == 0) ticks[0] = t1 - t0;
}

// the same 124 ordered additions as dependent v_add_f64 (operands already in registers)
__global__ void k_addchain(const double* a, double* d, unsigned long long* ticks) {
  const int l = threadIdx.x;
  double v[16];
  for (int k = 0; k < 16; k++) v[k] = a[64 * k + l];
  double acc = 0.0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int s = 0; s < 31; s++) {
    acc += v[(4 * s) & 15]; acc += v[(4 * s + 1) & 15]; acc += v[(4 * s + 2) & 15]; acc += v[(4 * s + 3) & 15];
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  d[l] = acc;
  if (l == 0) ticks[0] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  double *a, *b, *c, *d;
  unsigned long long* ticks;
  CK(hipHostMalloc(&a, 64 * 64 * 8));
  CK(hipHostMalloc(&b, 64 * 8));
  CK(hipHostMalloc(&c, 256 * 8));
  CK(hipHostMalloc(&d, 256 * 8));
  CK(hipHostMalloc(&ticks, 8));
  // ---- layout of the 4x4x4 (4 blocks) form: one-hot A lane x one-hot B lane -> which D lanes light up
  int a_of[64][3], b_of[64][3];  // block, i / j, k as deduced
  memset(a_of, -1, sizeof a_of);
  memset(b_of, -1, sizeof b_of);
  std::vector<std::vector<int>> hit(64 * 64);
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      for (int l = 0; l < 64; l++) { a[l] = l == la ? 1.0 : 0.0; b[l] = l == lb ? 1.0 : 0.0; c[l] = 0.0; }
      k_4x4x4<<<1, 64>>>(a, b, c, d);
      CK(hipDeviceSynchronize());
      for (int l = 0; l < 64; l++) if (d[l] != 0.0) hit[la * 64 + lb].push_back(l);
    }
  // print, per A lane, the B lanes it pairs with and the D lanes
  printf("4x4x4f64 layout: A lane -> (B lane -> D lanes)\n");
  for (int la = 0; la < 64; la++) {
    printf("A%2d:", la);
    for (int lb = 0; lb < 64; lb++) if (!hit[la * 64 + lb].empty()) { printf(" B%d->", lb); for (int x : hit[la * 64 + lb]) printf("D%d,", x); }
    printf("\n");
  }
  // ---- rounding order: the hypothesis  lane = 16 * block + 4 * k + i  for A (tested below by value, not assumed by the report)
  std::mt19937_64 rng(12345);
  auto rnd = [&]() {
    const double m = std::ldexp((double)(rng() >> 11), -53) + 0.5;
    const int e = (int)(rng() % 61) - 30;
    return ((rng() & 1) ? -m : m) * std::ldexp(1.0, e);
  };
  long n_seq = 0, n_rev = 0, n_wide = 0, n_other = 0, n_trials = 0;
  for (int t = 0; t < 20000; t++) {
    for (int l = 0; l < 64; l++) { a[l] = rnd(); b[l] = 1.0; c[l] = rnd(); }
    k_4x4x4<<<1, 64>>>(a, b, c, d);
    CK(hipDeviceSynchronize());
    // every D lane: find which four A lanes feed it from the one-hot table (B lane irrelevant: all ones)
    for (int ld = 0; ld < 64; ld++) {
      int src[4], ns = 0;
      for (int la = 0; la < 64 && ns < 4; la++) {
        bool feeds = false;
        for (int lb = 0; lb < 64 && !feeds; lb++) for (int x : hit[la * 64 + lb]) if (x == ld) { feeds = true; break; }
        if (feeds) src[ns++] = la;
      }
      if (ns != 4) continue;
      volatile double s1 = c[ld]; for (int k = 0; k < 4; k++) s1 = s1 + a[src[k]];
      volatile double s2 = c[ld]; for (int k = 3; k >= 0; k--) s2 = s2 + a[src[k]];
      long double w = c[ld]; for (int k = 0; k < 4; k++) w += a[src[k]];
      const double s3 = (double)w;
      n_trials++;
      if (d[ld] == s1) n_seq++;
      else if (d[ld] == s2) n_rev++;
      else if (d[ld] == s3) n_wide++;
      else n_other++;
    }
  }
  printf("4x4x4f64 rounding over %ld sums (A sources in ascending lane order): sequential-ascending %ld, sequential-descending %ld, wide-then-round %ld, other %ld\n",
         n_trials, n_seq, n_rev, n_wide, n_other);
  // ---- denormals and signed zeros
  {
    for (int l = 0; l < 64; l++) { a[l] = 0.0; b[l] = 1.0; c[l] = 0.0; }
    a[0] = 4.9406564584124654e-324; c[0] = 0.0;            // denormal addend
    k_4x4x4<<<1, 64>>>(a, b, c, d);
    CK(hipDeviceSynchronize());
    double mx = 0; for (int l = 0; l < 64; l++) mx = d[l] > mx ? d[l] : mx;
    printf("denormal addend kept: %s (max D = %g)\n", mx == 4.9406564584124654e-324 ? "yes" : "NO", mx);
    for (int l = 0; l < 64; l++) { a[l] = -0.0; b[l] = 1.0; c[l] = 0.0; }
    k_4x4x4<<<1, 64>>>(a, b, c, d);
    CK(hipDeviceSynchronize());
    int negz = 0; for (int l = 0; l < 64; l++) negz += std::signbit(d[l]) ? 1 : 0;
    printf("+0 + four -0 addends: %d of 64 results have the sign bit (IEEE sequential: 0)\n", negz);
    for (int l = 0; l < 64; l++) { a[l] = 1e308; b[l] = 1.0; c[l] = 1e308; }
    a[1] = a[5] = a[9] = a[13] = -1e308;
    k_4x4x4<<<1, 64>>>(a, b, c, d);
    CK(hipDeviceSynchronize());
    printf("overflow case D[0..3] = %g %g %g %g\n", d[0], d[1], d[2], d[3]);
  }
  // ---- 16x16x4 form, same question (layout known: A[l & 15][l >> 4], D row = (l >> 4) + 4 reg, col = l & 15)
  {
    long s_seq = 0, s_rev = 0, s_wide = 0, s_other = 0;
    for (int t = 0; t < 5000; t++) {
      for (int l = 0; l < 64; l++) { a[l] = rnd(); b[l] = 1.0; }
      for (int l = 0; l < 256; l++) c[l] = rnd();
      k_16x16x4<<<1, 64>>>(a, b, c, d);
      CK(hipDeviceSynchronize());
      for (int reg = 0; reg < 4; reg++)
        for (int l = 0; l < 64; l++) {
          const int row = (l >> 4) + 4 * reg;
          volatile double s1 = c[64 * reg + l]; for (int k = 0; k < 4; k++) s1 = s1 + a[row + 16 * k];
          volatile double s2 = c[64 * reg + l]; for (int k = 3; k >= 0; k--) s2 = s2 + a[row + 16 * k];
          long double w = c[64 * reg + l]; for (int k = 0; k < 4; k++) w += a[row + 16 * k];
          const double r = d[64 * reg + l];
          if (r == s1) s_seq++; else if (r == s2) s_rev++; else if (r == (double)w) s_wide++; else s_other++;
        }
    }
    printf("16x16x4f64 rounding: sequential k=0..3 %ld, sequential k=3..0 %ld, wide-then-round %ld, other %ld\n", s_seq, s_rev, s_wide, s_other);
  }
  // ---- latency of a dependent chain of 31 instructions (121 addends)
  for (int l = 0; l < 64 * 64; l++) a[l] = rnd();
  for (int rep = 0; rep < 3; rep++) {
    k_chain<<<1, 64>>>(a, 31, d, ticks);
    CK(hipDeviceSynchronize());
    printf("chain of 31 dependent 4x4x4f64: %llu s_memtime ticks\n", ticks[0]);
  }
  for (int rep = 0; rep < 3; rep++) {
    k_addchain<<<1, 64>>>(a, d, ticks);
    CK(hipDeviceSynchronize());
    printf("chain of 124 dependent v_add_f64: %llu s_memtime ticks\n", ticks[0]);
  }
  return 0;
}
