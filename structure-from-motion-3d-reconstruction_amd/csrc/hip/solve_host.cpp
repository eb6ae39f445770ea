// solve_host.cpp -- sfm::solve_gauss (cpp/include/dense.hpp:54-93) for the reduced camera system of a BA window (36 / 60
// unknowns), on the host core that is polling for the iteration's result anyway.  Compiled by g++ into libsfmx.so with
// -ffp-contract=off (no FMA: the reference's x86-64 build has none), so x is the reference's bit for bit.
//
// Why here: the system is built on the device (k_ba_points / k_ba_reduce) and is 10 KB; one wavefront needs ~30 us for its
// elimination (a dependent chain per pivot), a 4-5 GHz core 3-4 us, and the iteration is a host round trip either way
// (the SO(3) updates need the platform libm).  sfmx_solve_dense -- any n, pose graphs -- stays on the device.
#include <cmath>
#include <cstring>

// S [n][n] row-major, b [n]; work: n * (n + 1) doubles.  Returns 0, or 1 where the reference throws (pivot < 1e-15).
extern "C" int sfmx_host_solve_window(const double* S, const double* b, int n, double* x, double* work) {
  const int ld = n + 1;  // the right-hand side rides along as column n
  for (int i = 0; i < n; ++i) {
    std::memcpy(work + (size_t)i * ld, S + (size_t)i * n, (size_t)n * sizeof(double));
    work[(size_t)i * ld + n] = b[i];
  }
  for (int k = 0; k < n; ++k) {
    // partial pivoting: the FIRST strictly largest |a_ik|, i >= k (dense.hpp:61-66); a NaN never wins `v > best`
    int piv = k;
    double best = std::fabs(work[(size_t)k * ld + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(work[(size_t)i * ld + k]);
      if (v > best) { best = v; piv = i; }
    }
    if (best < 1e-15) return 1;  // dense.hpp:67
    double* rk = work + (size_t)k * ld;
    if (piv != k) {  // dense.hpp:69-72 (columns left of k are never read again)
      double* rp = work + (size_t)piv * ld;
      for (int j = k; j <= n; ++j) { const double t = rk[j]; rk[j] = rp[j]; rp[j] = t; }
    }
    const double akk = rk[k];
    for (int j = k; j <= n; ++j) rk[j] /= akk;  // dense.hpp:74-76
    for (int i = k + 1; i < n; ++i) {            // dense.hpp:78-83
      double* ri = work + (size_t)i * ld;
      const double f = ri[k];
      if (std::fabs(f) < 1e-18) continue;
      for (int j = k; j <= n; ++j) ri[j] -= f * rk[j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {  // dense.hpp:86-91
    const double* ri = work + (size_t)i * ld;
    double s = ri[n];
    for (int j = i + 1; j < n; ++j) s -= ri[j] * x[j];
    x[i] = s;
  }
  return 0;
}
