// host_math.hpp — the small, order-sensitive host-side math of the SfM hot path.
//
// These pieces stay on the host by design: they are O(1)..O(20) per call and
// use the platform libm (atan2/cos/sin/acos) exactly where the reference does, which is what makes
// the quantities that LEAVE the library (E of the winning hypothesis, R, t, triangulated points,
// pose updates) bit-identical to the reference.  Each function cites what it restates.
// Compile with -ffp-contract=off.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <functional>
#include <vector>

namespace sfmx_host {

struct V2 { double x = 0, y = 0; };
struct V3 { double x = 0, y = 0, z = 0; };
struct Mat3 {
  double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double& operator()(int r, int c) { return a[3 * r + c]; }
  double operator()(int r, int c) const { return a[3 * r + c]; }
  static Mat3 identity() { Mat3 m; m.a[0] = m.a[4] = m.a[8] = 1.0; return m; }
};

// cpp/include/linalg.hpp:37-88
inline V3 operator+(const V3& u, const V3& v) { return {u.x + v.x, u.y + v.y, u.z + v.z}; }
inline V3 operator-(const V3& u, const V3& v) { return {u.x - v.x, u.y - v.y, u.z - v.z}; }
inline V3 operator-(const V3& v) { return {-v.x, -v.y, -v.z}; }
inline V3 operator*(double s, const V3& v) { return {s * v.x, s * v.y, s * v.z}; }
inline double dot(const V3& u, const V3& v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline V3 cross(const V3& u, const V3& v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
inline double norm(const V3& v) { return std::sqrt(dot(v, v)); }
inline V3 unit(const V3& v) {
  const double n = norm(v);
  if (!std::isfinite(n) || n < 1e-12) return {0, 0, 0};
  return {v.x / n, v.y / n, v.z / n};
}
inline Mat3 operator*(const Mat3& A, const Mat3& B) {
  Mat3 C;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A(r, k) * B(k, c);
      C(r, c) = s;
    }
  return C;
}
inline V3 operator*(const Mat3& A, const V3& v) {
  return {A(0, 0) * v.x + A(0, 1) * v.y + A(0, 2) * v.z, A(1, 0) * v.x + A(1, 1) * v.y + A(1, 2) * v.z,
          A(2, 0) * v.x + A(2, 1) * v.y + A(2, 2) * v.z};
}
inline Mat3 transpose(const Mat3& A) {
  Mat3 T;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T(r, c) = A(c, r);
  return T;
}
inline double det(const Mat3& A) {
  return A(0, 0) * (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) - A(0, 1) * (A(1, 0) * A(2, 2) - A(1, 2) * A(2, 0)) +
         A(0, 2) * (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0));
}

// linalg.hpp:90-125
inline Mat3 so3_exp(const V3& w) {
  const double th = norm(w);
  Mat3 R = Mat3::identity();
  if (th < 1e-10) {
    R(0, 1) = -w.z; R(0, 2) = w.y; R(1, 0) = w.z; R(1, 2) = -w.x; R(2, 0) = -w.y; R(2, 1) = w.x;
    return R;
  }
  const double ax = w.x / th, ay = w.y / th, az = w.z / th;
  const double c = std::cos(th), s = std::sin(th), C = 1 - c;
  R(0, 0) = c + ax * ax * C;      R(0, 1) = ax * ay * C - az * s; R(0, 2) = ax * az * C + ay * s;
  R(1, 0) = ay * ax * C + az * s; R(1, 1) = c + ay * ay * C;      R(1, 2) = ay * az * C - ax * s;
  R(2, 0) = az * ax * C - ay * s; R(2, 1) = az * ay * C + ax * s; R(2, 2) = c + az * az * C;
  return R;
}
inline V3 so3_log(const Mat3& R) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  double ct = (tr - 1.0) * 0.5;
  ct = std::max(-1.0, std::min(1.0, ct));
  const double th = std::acos(ct);
  if (th < 1e-10) return {0, 0, 0};
  const double s = std::sin(th);
  const double k = th / (2.0 * s);
  return {k * (R(2, 1) - R(1, 2)), k * (R(0, 2) - R(2, 0)), k * (R(1, 0) - R(0, 1))};
}

// linalg.hpp:133-201 (classical Jacobi with the platform libm)
inline void jacobi_eig(const double* Ain, int N, int sweeps, double* w_out, double* V_out) {
  double A[81], V[81], w[9];
  int perm[9];
  for (int i = 0; i < N * N; i++) { A[i] = Ain[i]; V[i] = 0.0; }
  for (int i = 0; i < N; i++) V[i * N + i] = 1.0;
  for (int it = 0; it < sweeps; ++it) {
    int p = 0, q = 1;
    double big = 0;
    for (int i = 0; i < N; i++)
      for (int j = i + 1; j < N; j++) {
        const double v = std::fabs(A[i * N + j]);
        if (v > big) { big = v; p = i; q = j; }
      }
    if (big < 1e-12) break;
    const double phi = 0.5 * std::atan2(2.0 * A[p * N + q], (A[q * N + q] - A[p * N + p]));
    const double c = std::cos(phi), s = std::sin(phi);
    for (int k = 0; k < N; k++) {
      const double ap = A[p * N + k], aq = A[q * N + k];
      A[p * N + k] = c * ap - s * aq;
      A[q * N + k] = s * ap + c * aq;
    }
    for (int k = 0; k < N; k++) {
      const double ap = A[k * N + p], aq = A[k * N + q];
      A[k * N + p] = c * ap - s * aq;
      A[k * N + q] = s * ap + c * aq;
    }
    A[p * N + q] = 0.0;
    A[q * N + p] = 0.0;
    for (int k = 0; k < N; k++) {
      const double vp = V[k * N + p], vq = V[k * N + q];
      V[k * N + p] = c * vp - s * vq;
      V[k * N + q] = s * vp + c * vq;
    }
  }
  for (int i = 0; i < N; i++) { w[i] = A[i * N + i]; perm[i] = i; }
  // std::sort on <= 16 elements == libstdc++ __insertion_sort
  for (int i = 1; i < N; i++) {
    const int val = perm[i];
    if (w[val] < w[perm[0]]) {
      for (int k = i; k > 0; k--) perm[k] = perm[k - 1];
      perm[0] = val;
    } else {
      int k = i;
      while (w[val] < w[perm[k - 1]]) { perm[k] = perm[k - 1]; k--; }
      perm[k] = val;
    }
  }
  for (int c = 0; c < N; c++) {
    w_out[c] = w[perm[c]];
    for (int r = 0; r < N; r++) V_out[r * N + c] = V[r * N + perm[c]];
  }
}

// T:503-517
inline void ata_upper(const double* A, int rows, int cols, double* M) {
  for (int i = 0; i < cols; i++)
    for (int j = i; j < cols; j++) {
      double s = 0;
      for (int r = 0; r < rows; r++) s += A[r * cols + i] * A[r * cols + j];
      M[i * cols + j] = s;
      M[j * cols + i] = s;
    }
}

struct Svd { Mat3 U; double s[3]; Mat3 V; };
// T:537-593
inline Svd svd3(const Mat3& A) {
  double G[9], w[3], Ve[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A(k, r) * A(k, c);
      G[3 * r + c] = s;
    }
  jacobi_eig(G, 3, 80, w, Ve);
  double sv[3];
  for (int i = 0; i < 3; i++) sv[i] = std::sqrt(std::max(0.0, w[i]));
  int ord[3] = {0, 1, 2};
  for (int i = 1; i < 3; i++) {
    const int val = ord[i];
    if (sv[val] > sv[ord[0]]) {
      for (int k = i; k > 0; k--) ord[k] = ord[k - 1];
      ord[0] = val;
    } else {
      int k = i;
      while (sv[val] > sv[ord[k - 1]]) { ord[k] = ord[k - 1]; k--; }
      ord[k] = val;
    }
  }
  Svd o;
  for (int c = 0; c < 3; c++) {
    o.s[c] = sv[ord[c]];
    for (int r = 0; r < 3; r++) o.V(r, c) = Ve[3 * r + ord[c]];
  }
  V3 u[3];
  for (int c = 0; c < 3; c++) {
    V3 t = A * V3{o.V(0, c), o.V(1, c), o.V(2, c)};
    if (o.s[c] > 1e-12) t = {t.x / o.s[c], t.y / o.s[c], t.z / o.s[c]};
    else t = unit(t);
    u[c] = t;
  }
  u[0] = unit(u[0]);
  u[1] = u[1] - dot(u[0], u[1]) * u[0];
  u[1] = unit(u[1]);
  u[2] = unit(cross(u[0], u[1]));
  for (int c = 0; c < 3; c++) { o.U(0, c) = u[c].x; o.U(1, c) = u[c].y; o.U(2, c) = u[c].z; }
  return o;
}

// T:595-627
inline Mat3 eight_point_E(const double* xn, const double* yn, const int* idx8) {
  double A[72], G[81], w[9], V[81];
  for (int r = 0; r < 8; r++) {
    const int i = idx8[r];
    const double x = xn[2 * i], y = xn[2 * i + 1], xp = yn[2 * i], yp = yn[2 * i + 1];
    double* row = A + 9 * r;
    row[0] = xp * x; row[1] = xp * y; row[2] = xp; row[3] = yp * x; row[4] = yp * y; row[5] = yp;
    row[6] = x; row[7] = y; row[8] = 1.0;
  }
  ata_upper(A, 8, 9, G);
  jacobi_eig(G, 9, 120, w, V);
  Mat3 E;
  for (int r = 0; r < 9; r++) E.a[r] = V[r * 9];
  const Svd d = svd3(E);
  Mat3 S;
  S(0, 0) = d.s[0]; S(1, 1) = d.s[1]; S(2, 2) = 0.0;
  return (d.U * S) * transpose(d.V);
}

// T:471-501; false where the reference throws "Singular K"
inline bool invert_K(const Mat3& K, Mat3& inv) {
  const double d = det(K);
  if (std::fabs(d) < 1e-12) return false;
  inv(0, 0) = (K(1, 1) * K(2, 2) - K(1, 2) * K(2, 1)) / d;
  inv(0, 1) = -(K(0, 1) * K(2, 2) - K(0, 2) * K(2, 1)) / d;
  inv(0, 2) = (K(0, 1) * K(1, 2) - K(0, 2) * K(1, 1)) / d;
  inv(1, 0) = -(K(1, 0) * K(2, 2) - K(1, 2) * K(2, 0)) / d;
  inv(1, 1) = (K(0, 0) * K(2, 2) - K(0, 2) * K(2, 0)) / d;
  inv(1, 2) = -(K(0, 0) * K(1, 2) - K(0, 2) * K(1, 0)) / d;
  inv(2, 0) = (K(1, 0) * K(2, 1) - K(1, 1) * K(2, 0)) / d;
  inv(2, 1) = -(K(0, 0) * K(2, 1) - K(0, 1) * K(2, 0)) / d;
  inv(2, 2) = (K(0, 0) * K(1, 1) - K(0, 1) * K(1, 0)) / d;
  return true;
}
inline V2 norm_point(const Mat3& Kinv, const V2& p) {
  const V3 h = Kinv * V3{p.x, p.y, 1.0};
  return {h.x / h.z, h.y / h.z};
}

// std::mt19937 + libstdc++ 11 uniform_int_distribution<int> (Lemire), as drawn at T:657-665
struct Mt19937 {
  std::uint32_t s[624];
  int pos;
  explicit Mt19937(std::uint32_t seed) {
    s[0] = seed;
    for (int i = 1; i < 624; i++) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (std::uint32_t)i;
    pos = 624;
  }
  std::uint32_t next() {
    if (pos >= 624) {
      for (int i = 0; i < 624; i++) {
        const std::uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
        s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    std::uint32_t y = s[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
  int below(std::uint32_t range) {
    std::uint64_t prod = (std::uint64_t)next() * range;
    std::uint32_t low = (std::uint32_t)prod;
    if (low < range) {
      const std::uint32_t thr = (0u - range) % range;
      while (low < thr) {
        prod = (std::uint64_t)next() * range;
        low = (std::uint32_t)prod;
      }
    }
    return (int)(prod >> 32);
  }
};

// camera->world pose (T:157-178)
struct Pose {
  Mat3 R = Mat3::identity();
  V3 t{0, 0, 0};
};
inline void inv_wc(const Pose& p, Mat3& Rwc, V3& twc) {
  Rwc = transpose(p.R);
  twc = -(Rwc * p.t);
}
inline Pose compose_right_inv(const Pose& cur, const Mat3& R_ji, const V3& t_ji) {
  const Mat3 Rd = transpose(R_ji);
  const V3 td = -(transpose(R_ji) * t_ji);
  Pose o;
  o.R = cur.R * Rd;
  o.t = (cur.R * td) + cur.t;
  return o;
}

// DLT on 4 rows -> smallest eigenvector / w (T:719-727, T:1510-1515)
inline V3 dlt_solve(const double* A16) {
  double G[16], w[4], V[16];
  ata_upper(A16, 4, 4, G);
  jacobi_eig(G, 4, 80, w, V);
  const double ww = V[12];
  return {V[0] / ww, V[4] / ww, V[8] / ww};
}
// T:1477-1516
inline bool triangulate_dlt(const Mat3& K, const Pose& pi, const Pose& pj, V2 ui, V2 uj, V3& X) {
  Mat3 Ri, Rj, Kinv;
  V3 ti, tj;
  inv_wc(pi, Ri, ti);
  inv_wc(pj, Rj, tj);
  if (!invert_K(K, Kinv)) return false;
  const V2 a = norm_point(Kinv, ui), b = norm_point(Kinv, uj);
  const double A[16] = {a.x * Ri(2, 0) - Ri(0, 0), a.x * Ri(2, 1) - Ri(0, 1), a.x * Ri(2, 2) - Ri(0, 2), a.x * ti.z - ti.x,
                        a.y * Ri(2, 0) - Ri(1, 0), a.y * Ri(2, 1) - Ri(1, 1), a.y * Ri(2, 2) - Ri(1, 2), a.y * ti.z - ti.y,
                        b.x * Rj(2, 0) - Rj(0, 0), b.x * Rj(2, 1) - Rj(0, 1), b.x * Rj(2, 2) - Rj(0, 2), b.x * tj.z - tj.x,
                        b.y * Rj(2, 0) - Rj(1, 0), b.y * Rj(2, 1) - Rj(1, 1), b.y * Rj(2, 2) - Rj(1, 2), b.y * tj.z - tj.y};
  X = dlt_solve(A);
  return true;
}

// E -> (R,t) with the 4-candidate cheirality vote on the first min(20,n) inliers (T:680-760)
using ParallelFor = std::function<void(int, const std::function<void(int)>&)>;
inline void decompose_E(const Mat3& E, const double* xi, const double* xj, const std::vector<int>& inl, Mat3& R_out, V3& t_out,
                        int* cand_out = nullptr, const ParallelFor& par = nullptr) {
  const Svd d = svd3(E);
  Mat3 W;
  W(0, 1) = -1; W(1, 0) = 1; W(2, 2) = 1;
  const Mat3 Vt = transpose(d.V);
  Mat3 R1 = d.U * W * Vt;
  Mat3 R2 = d.U * transpose(W) * Vt;
  if (det(R1) < 0) for (double& v : R1.a) v = -v;
  if (det(R2) < 0) for (double& v : R2.a) v = -v;
  const V3 t = unit(V3{d.U(0, 2), d.U(1, 2), d.U(2, 2)});
  const Mat3 Rc[4] = {R1, R1, R2, R2};
  const V3 tc[4] = {t, V3{-t.x, -t.y, -t.z}, t, V3{-t.x, -t.y, -t.z}};
  // 4 candidates x min(20,n) independent DLT solves; `par` (optional) runs them on a pool.  The vote is
  // a sum of 0/1 flags, so the result does not depend on the execution order.
  const int M = std::min((int)inl.size(), 20);
  std::vector<unsigned char> pass((size_t)4 * (size_t)std::max(M, 1), 0);
  auto one = [&](int job) {
    const int c = job / M, k = job % M;
    const int i = inl[(size_t)k];
    const V2 a{xi[2 * i], xi[2 * i + 1]}, b{xj[2 * i], xj[2 * i + 1]};
    const Mat3& R = Rc[c];
    const V3& tt = tc[c];
    const double A[16] = {-1, 0, a.x, 0, 0, -1, a.y, 0,
                          b.x * R(2, 0) - R(0, 0), b.x * R(2, 1) - R(0, 1), b.x * R(2, 2) - R(0, 2), b.x * tt.z - tt.x,
                          b.y * R(2, 0) - R(1, 0), b.y * R(2, 1) - R(1, 1), b.y * R(2, 2) - R(1, 2), b.y * tt.z - tt.y};
    const V3 X = dlt_solve(A);
    const V3 X2 = (R * X) + tt;
    pass[(size_t)job] = (X.z > 0 && X2.z > 0) ? 1 : 0;
  };
  if (M > 0) {
    if (par) par(4 * M, one);
    else for (int job = 0; job < 4 * M; job++) one(job);
  }
  int best = 0, bestok = -1;
  for (int c = 0; c < 4; c++) {
    int ok = 0;
    for (int k = 0; k < M; k++) ok += pass[(size_t)(c * M + k)];
    if (ok > bestok) { bestok = ok; best = c; }
  }
  R_out = Rc[best];
  t_out = tc[best];
  if (cand_out) *cand_out = best;
}

}  // namespace sfmx_host
