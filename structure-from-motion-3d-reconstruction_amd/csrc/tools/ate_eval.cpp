// ate_keyframes — absolute trajectory error of a keyframe CSV against Middlebury ground truth.
//
// The metric of BASELINE.json ("ATE-RMSE vs Middlebury GT") is *defined* by the reference's evaluator
// (cpp/tools/ate_keyframes.cpp): camera centres C = -R^T t from <name>_par.txt (:193-196), Umeyama
// alignment of the estimated centres onto them with or without scale (:334-389), RMSE / mean / median /
// max of the residual norms, printed as a fixed text block (:440-476).  This is the build's own host tool
// for stating that metric without shipping reference code: same command line, same stdout block and exit
// codes, and -- because every sum, product and libm call is evaluated in the same order -- the same
// digits (tests/test_tools.py compares with the output of the real tool on committed fixtures).
// It is an evaluator, not part of the hot path: plain host C++, no device code.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "../host/host_math.hpp"

namespace {
using sfmx_host::Mat3;
using sfmx_host::V3;

V3 scaled(const V3& v, double s) { return {s * v.x, s * v.y, s * v.z}; }
V3 divided(const V3& v, double s) { return {v.x / s, v.y / s, v.z / s}; }
V3 column(const Mat3& M, int j) { return {M(0, j), M(1, j), M(2, j)}; }
void put_column(Mat3& M, int j, const V3& v) { M(0, j) = v.x; M(1, j) = v.y; M(2, j) = v.z; }

struct Options {
  std::string par, csv;
  int start = 0, count = 4;
  bool with_scale = true;
  bool ok() const { return !par.empty() && !csv.empty() && count > 1 && start >= 0; }
};

// "--flag value" pairs may appear anywhere; an unparsable integer leaves the default (reference :41-74)
Options read_options(int argc, char** argv) {
  Options o;
  auto whole_int = [](const std::string& s, int& out) {
    try {
      size_t used = 0;
      const int v = std::stoi(s, &used);
      if (used == s.size()) out = v;
    } catch (...) {
    }
  };
  // the reference looks each flag up independently and takes its FIRST occurrence
  bool got_par = false, got_csv = false, got_start = false, got_count = false;
  for (int i = 0; i + 1 < argc; ++i) {
    const std::string a = argv[i], v = argv[i + 1];
    if (a == "--par" && !got_par) { o.par = v; got_par = true; }
    else if (a == "--keyframes" && !got_csv) { o.csv = v; got_csv = true; }
    else if (a == "--start" && !got_start) { whole_int(v, o.start); got_start = true; }
    else if (a == "--count" && !got_count) { whole_int(v, o.count); got_count = true; }
  }
  bool se3 = false, sim3 = false;
  for (int i = 0; i < argc; ++i) {
    const std::string a = argv[i];
    se3 |= a == "--se3";
    sim3 |= a == "--sim3";
  }
  if (se3) o.with_scale = false;
  if (sim3) o.with_scale = true;  // --sim3 wins when both are given
  return o;
}

void print_usage() {
  std::cerr << "ate_keyframes (C++20, no OpenCV)\n"
            << "Compute ATE RMSE over N keyframes using ground-truth poses from Middlebury *_par.txt.\n\n"
            << "Usage:\n"
            << "  ate_keyframes --par <templeR_par.txt> --keyframes <keyframes_camera_centers.csv>\n"
            << "               [--start 0 --count 4] [--sim3|--se3]\n\n"
            << "Notes:\n"
            << "  - --sim3 (default) uses similarity alignment (scale + rotation + translation), typical for monocular.\n"
            << "  - --se3 uses rigid alignment (rotation + translation only).\n";
}

// one CSV record; double quotes toggle "inside a quoted field" and are dropped
std::vector<std::string> csv_fields(const std::string& line) {
  std::vector<std::string> f(1);
  bool quoted = false;
  for (char ch : line) {
    if (ch == '"') quoted = !quoted;
    else if (ch == ',' && !quoted) f.emplace_back();
    else f.back().push_back(ch);
  }
  return f;
}

struct Row { std::string image; V3 centre; };

// rows whose field count differs from the header's, or whose x/y/z do not parse, are skipped (:121-152)
std::vector<Row> load_keyframes(const std::string& path) {
  std::vector<Row> rows;
  std::ifstream in(path);
  std::string line;
  if (!in || !std::getline(in, line)) return rows;
  const std::vector<std::string> head = csv_fields(line);
  int ci = -1, cx = -1, cy = -1, cz = -1;
  for (int i = (int)head.size() - 1; i >= 0; --i) {  // first match wins
    if (head[(size_t)i] == "image") ci = i;
    if (head[(size_t)i] == "x") cx = i;
    if (head[(size_t)i] == "y") cy = i;
    if (head[(size_t)i] == "z") cz = i;
  }
  if (ci < 0 || cx < 0 || cy < 0 || cz < 0) return rows;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> f = csv_fields(line);
    if (f.size() != head.size()) continue;
    try {
      Row r;
      r.image = f[(size_t)ci];
      r.centre = {std::stod(f[(size_t)cx]), std::stod(f[(size_t)cy]), std::stod(f[(size_t)cz])};
      rows.push_back(std::move(r));
    } catch (...) {
    }
  }
  return rows;
}

struct GtPose { Mat3 R; V3 t; };

// "<name> k11..k33 r11..r33 t1 t2 t3" per line after the count line; the first record of a name is kept (:159-191)
bool load_par(const std::string& path, std::map<std::string, GtPose>& out) {
  std::ifstream in(path);
  std::string line;
  if (!in || !std::getline(in, line)) return false;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    std::istringstream ss(line);
    std::string name;
    ss >> name;
    if (name.empty()) continue;
    double v[21];
    int got = 0;
    while (got < 21 && (ss >> v[got])) ++got;
    if (got < 21) continue;
    GtPose g;
    for (int i = 0; i < 9; i++) g.R.a[i] = v[9 + i];
    g.t = {v[18], v[19], v[20]};
    out.emplace(name, g);
  }
  return true;
}

// symmetric 3x3 eigen-decomposition as the evaluator defines it (:205-259): <= 64 classical Jacobi rotations,
// stop below 1e-15, symmetrised pivot entry after each rotation; eigenvectors in the columns of V
void eig_sym3(Mat3 A, Mat3& V, double lam[3]) {
  V = Mat3::identity();
  for (int sweep = 0; sweep < 64; ++sweep) {
    int p = 0, q = 1;
    double big = std::fabs(A(0, 1));
    if (std::fabs(A(0, 2)) > big) { big = std::fabs(A(0, 2)); p = 0; q = 2; }
    if (std::fabs(A(1, 2)) > big) { big = std::fabs(A(1, 2)); p = 1; q = 2; }
    if (big < 1e-15) break;
    const double phi = 0.5 * std::atan2(2.0 * A(p, q), A(q, q) - A(p, p));
    const double c = std::cos(phi), s = std::sin(phi);
    for (int k = 0; k < 3; ++k) {  // rows p,q
      const double u = A(p, k), w = A(q, k);
      A(p, k) = c * u - s * w;
      A(q, k) = s * u + c * w;
    }
    for (int k = 0; k < 3; ++k) {  // columns p,q
      const double u = A(k, p), w = A(k, q);
      A(k, p) = c * u - s * w;
      A(k, q) = s * u + c * w;
    }
    A(p, q) = A(q, p) = 0.5 * (A(p, q) + A(q, p));
    for (int k = 0; k < 3; ++k) {
      const double u = V(k, p), w = V(k, q);
      V(k, p) = c * u - s * w;
      V(k, q) = s * u + c * w;
    }
  }
  lam[0] = A(0, 0); lam[1] = A(1, 1); lam[2] = A(2, 2);
}

// least-squares similarity (or rigid) transform dst ~ s R src + t, Umeyama 1991 as evaluated at :334-389
class Similarity {
 public:
  Similarity(const std::vector<V3>& src, const std::vector<V3>& dst, bool with_scale) {
    const size_t n = src.size();
    const double dn = (double)n;
    V3 ms, md;
    for (size_t i = 0; i < n; ++i) { ms = ms + src[i]; md = md + dst[i]; }
    ms = divided(ms, dn);
    md = divided(md, dn);
    std::vector<V3> x(n), y(n);
    for (size_t i = 0; i < n; ++i) { x[i] = src[i] - ms; y[i] = dst[i] - md; }
    Mat3 cov;  // (1/N) sum y x^T
    for (size_t i = 0; i < n; ++i) {
      const double yy[3] = {y[i].x, y[i].y, y[i].z}, xx[3] = {x[i].x, x[i].y, x[i].z};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) cov(r, c) += yy[r] * xx[c];
    }
    const double inv_n = 1.0 / dn;
    for (double& e : cov.a) e *= inv_n;
    Mat3 U, V;
    double sv[3];
    svd(cov, U, V, sv);
    Mat3 D = Mat3::identity();
    if (sfmx_host::det(U) * sfmx_host::det(V) < 0.0) D(2, 2) = -1.0;
    R_ = U * D * sfmx_host::transpose(V);
    double var = 0.0;
    for (size_t i = 0; i < n; ++i) var += sfmx_host::dot(x[i], x[i]);
    var *= inv_n;
    double s = 1.0;
    if (with_scale) {
      const double tr = sv[0] * D(0, 0) + sv[1] * D(1, 1) + sv[2] * D(2, 2);
      if (var > 1e-15) s = tr / var;
    }
    t_ = md - scaled(R_ * ms, s);
    s_ = with_scale ? s : 1.0;
  }
  V3 operator()(const V3& p) const { return scaled(R_ * p, s_) + t_; }
  double scale() const { return s_; }

 private:
  // SVD through the eigen-decomposition of M^T M (:279-326): singular values descending, u_i = M v_i / s_i
  // (zero vector when s_i < 1e-12), Gram-Schmidt with fixed fall-back axes, u2 = u0 x u1
  static void svd(const Mat3& M, Mat3& U, Mat3& V, double sv[3]) {
    Mat3 Ev;
    double lam[3];
    eig_sym3(sfmx_host::transpose(M) * M, Ev, lam);
    struct Item { int idx; double val; };
    Item it[3];
    for (int i = 0; i < 3; ++i) it[i] = {i, std::sqrt(std::max(0.0, lam[i]))};
    std::sort(it, it + 3, [](const Item& a, const Item& b) { return a.val > b.val; });
    V3 u[3];
    for (int j = 0; j < 3; ++j) {
      put_column(V, j, column(Ev, it[j].idx));
      sv[j] = it[j].val;
      u[j] = sv[j] < 1e-12 ? V3{0, 0, 0} : divided(M * column(V, j), sv[j]);
    }
    using sfmx_host::cross; using sfmx_host::norm; using sfmx_host::unit;
    u[0] = norm(u[0]) > 1e-12 ? unit(u[0]) : V3{1, 0, 0};
    u[1] = u[1] - scaled(u[0], sfmx_host::dot(u[1], u[0]));
    u[1] = norm(u[1]) > 1e-12 ? unit(u[1]) : unit(cross(u[0], V3{0, 0, 1}));
    if (norm(u[1]) < 1e-12) u[1] = unit(cross(u[0], V3{0, 1, 0}));
    u[2] = unit(cross(u[0], u[1]));
    for (int j = 0; j < 3; ++j) put_column(U, j, u[j]);
  }
  Mat3 R_;
  V3 t_;
  double s_ = 1.0;
};
}  // namespace

int main(int argc, char** argv) {
  const Options opt = read_options(argc, argv);
  if (!opt.ok()) {
    print_usage();
    return 2;
  }
  const std::vector<Row> rows = load_keyframes(opt.csv);
  if (rows.empty()) {
    std::cerr << "Failed to read keyframes CSV or missing columns: " << opt.csv << "\n";
    return 2;
  }
  if (opt.start + opt.count > (int)rows.size()) {
    std::cerr << "Requested range exceeds keyframes CSV rows: start=" << opt.start << " count=" << opt.count << " rows=" << rows.size() << "\n";
    return 2;
  }
  std::map<std::string, GtPose> gt_of;
  if (!load_par(opt.par, gt_of)) {
    std::cerr << "Failed to read par file: " << opt.par << "\n";
    return 2;
  }
  std::vector<V3> est, gt;
  std::vector<std::string> names;
  for (int k = 0; k < opt.count; ++k) {
    const Row& r = rows[(size_t)(opt.start + k)];
    const auto it = gt_of.find(r.image);
    if (it == gt_of.end()) {
      std::cerr << "Image name not found in par file: " << r.image << "\n";
      return 2;
    }
    est.push_back(r.centre);
    gt.push_back(scaled(sfmx_host::transpose(it->second.R) * it->second.t, -1.0));  // C = -R^T t
    names.push_back(r.image);
  }
  const Similarity align(est, gt, opt.with_scale);
  std::vector<double> err(est.size());
  double mse = 0.0;
  for (size_t i = 0; i < est.size(); ++i) {
    err[i] = sfmx_host::norm(align(est[i]) - gt[i]);
    mse += err[i] * err[i];
  }
  mse /= (double)err.size();
  std::vector<double> sorted = err;
  std::sort(sorted.begin(), sorted.end());
  const double mean = std::accumulate(err.begin(), err.end(), 0.0) / (double)err.size();

  std::cout << "ATE (N keyframes)\n"
            << "  mode: " << (opt.with_scale ? "Sim(3)" : "SE(3)") << "\n"
            << "  start: " << opt.start << "  count: " << opt.count << "\n"
            << "  keyframes:\n";
  for (size_t i = 0; i < names.size(); ++i) std::cout << "    [" << (opt.start + (int)i) << "] " << names[i] << "\n";
  if (opt.with_scale) std::cout << "  scale (s): " << align.scale() << "\n";
  std::cout << "  ATE_RMSE: " << std::sqrt(mse) << "\n"
            << "  mean/median/max: " << mean << " / " << sorted[sorted.size() / 2] << " / " << sorted.back() << "\n"
            << "  per_frame_error:\n";
  for (size_t i = 0; i < names.size(); ++i) std::cout << "    " << names[i] << ": " << err[i] << "\n";
  return 0;
}
