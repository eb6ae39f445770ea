// ate_two_frames — trajectory error of TWO keyframes against Middlebury ground truth (BASELINE config 0:
// "TempleRing 2-keyframe pair ... ate_two_frames check").  With two camera centres the alignment is closed form: the
// minimal rotation taking the estimated baseline direction onto the true one, the length ratio as scale (Sim(3)) and
// the translation that pins the first centre (reference cpp/tools/ate_two_frames.cpp:243-306); the residual then
// sits entirely on the second keyframe.  Same command line, stdout block (scientific, 12 digits) and exit codes as the
// reference tool; tests/test_tools.py compares with the real tool's output.  Host-only evaluator, no device code.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../host/host_math.hpp"

namespace {
using sfmx_host::Mat3;
using sfmx_host::V3;
using sfmx_host::cross;
using sfmx_host::dot;
using sfmx_host::norm;
using sfmx_host::unit;

struct Options { std::string par, csv; int i = 0, j = 1; bool with_scale = true; };

// first occurrence of a flag decides; a flag in last position has no value (reference :32-40)
bool flag_value(int argc, char** argv, const std::string& flag, std::string& out) {
  for (int k = 0; k < argc; ++k)
    if (flag == argv[k]) {
      if (k + 1 >= argc) return false;
      out = argv[k + 1];
      return true;
    }
  return false;
}
void whole_int(const std::string& s, int& out) {
  try {
    size_t used = 0;
    const int v = std::stoi(s, &used);
    if (used == s.size()) out = v;
  } catch (...) {
  }
}
Options read_options(int argc, char** argv) {
  Options o;
  std::string v;
  if (flag_value(argc, argv, "--par", v)) o.par = v;
  if (flag_value(argc, argv, "--keyframes", v)) o.csv = v;
  if (flag_value(argc, argv, "--i", v)) whole_int(v, o.i);
  if (flag_value(argc, argv, "--j", v)) whole_int(v, o.j);
  bool se3 = false, sim3 = false;
  for (int k = 0; k < argc; ++k) {
    se3 |= std::string(argv[k]) == "--se3";
    sim3 |= std::string(argv[k]) == "--sim3";
  }
  if (se3) o.with_scale = false;
  if (sim3) o.with_scale = true;
  return o;
}

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

// a row needs enough fields to reach the image/x/y/z columns (extra or missing trailing fields are fine, :141)
bool load_keyframes(const std::string& path, std::vector<Row>& rows) {
  std::ifstream in(path);
  std::string line;
  if (!in || !std::getline(in, line)) return false;
  const std::vector<std::string> head = csv_fields(line);
  int ci = -1, cx = -1, cy = -1, cz = -1;
  for (int k = (int)head.size() - 1; k >= 0; --k) {  // first match wins
    if (head[(size_t)k] == "image") ci = k;
    if (head[(size_t)k] == "x") cx = k;
    if (head[(size_t)k] == "y") cy = k;
    if (head[(size_t)k] == "z") cz = k;
  }
  if (ci < 0 || cx < 0 || cy < 0 || cz < 0) return false;
  const size_t need = (size_t)std::max(std::max(ci, cx), std::max(cy, cz));
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> f = csv_fields(line);
    if (f.size() <= need) continue;
    try {
      Row r;
      r.image = f[(size_t)ci];
      r.centre = {std::stod(f[(size_t)cx]), std::stod(f[(size_t)cy]), std::stod(f[(size_t)cz])};
      rows.push_back(std::move(r));
    } catch (...) {
    }
  }
  return true;
}

struct GtPose { Mat3 R; V3 t; };

// any record that is not "<name> + 21 numbers" makes the whole file unreadable (:175-183)
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
    for (double& e : v)
      if (!(ss >> e)) return false;
    GtPose g;
    for (int k = 0; k < 9; k++) g.R.a[k] = v[9 + k];
    g.t = {v[18], v[19], v[20]};
    out.emplace(name, g);
  }
  return true;
}

Mat3 scaled(const Mat3& A, double s) { Mat3 C; for (int k = 0; k < 9; k++) C.a[k] = A.a[k] * s; return C; }
Mat3 sum(const Mat3& A, const Mat3& B) { Mat3 C; for (int k = 0; k < 9; k++) C.a[k] = A.a[k] + B.a[k]; return C; }
Mat3 outer(const V3& u, const V3& v) {
  Mat3 M;
  const double uu[3] = {u.x, u.y, u.z}, vv[3] = {v.x, v.y, v.z};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) M(r, c) = uu[r] * vv[c];
  return M;
}

// minimal rotation with R a = b for the directions of a and b (:243-281): Rodrigues about a x b; identity for parallel
// vectors, a half turn about an axis orthogonal to a for anti-parallel ones
Mat3 rotation_between(const V3& a_raw, const V3& b_raw) {
  const V3 a = unit(a_raw), b = unit(b_raw);
  const double c = dot(a, b);
  const V3 v = cross(a, b);
  const double s = norm(v);
  const Mat3 I = Mat3::identity();
  if (s < 1e-12) {
    if (c > 0.0) return I;
    V3 axis;
    if (std::fabs(a.x) < std::fabs(a.y) && std::fabs(a.x) < std::fabs(a.z)) axis = {1, 0, 0};
    else if (std::fabs(a.y) < std::fabs(a.z)) axis = {0, 1, 0};
    else axis = {0, 0, 1};
    axis = unit(cross(a, axis));
    return sum(scaled(outer(axis, axis), 2.0), scaled(I, -1.0));
  }
  const V3 k = {v.x / s, v.y / s, v.z / s};
  const double angle = std::atan2(s, c), ca = std::cos(angle), sa = std::sin(angle);
  Mat3 Kx;
  Kx(0, 1) = -k.z; Kx(0, 2) = k.y; Kx(1, 0) = k.z; Kx(1, 2) = -k.x; Kx(2, 0) = -k.y; Kx(2, 1) = k.x;
  return sum(sum(scaled(I, ca), scaled(outer(k, k), 1.0 - ca)), scaled(Kx, sa));
}
V3 times(const V3& v, double s) { return {s * v.x, s * v.y, s * v.z}; }
}  // namespace

int main(int argc, char** argv) {
  const Options opt = read_options(argc, argv);
  if (opt.par.empty() || opt.csv.empty()) {
    std::cerr << "ate_two_frames (C++20, no OpenCV)\n"
              << "Compute ATE RMSE for two keyframes using ground-truth poses from Middlebury *_par.txt.\n\n"
              << "Usage:\n"
              << "  ate_two_frames --par <templeR_par.txt> --keyframes <keyframes_camera_centers.csv> [--i 0 --j 1] [--sim3|--se3]\n\n"
              << "Notes:\n"
              << "  - --sim3 (default) uses similarity alignment (scale + rotation + translation), typical for monocular.\n"
              << "  - --se3 uses rigid alignment (rotation + translation only).\n";
    return 2;
  }
  if (opt.i < 0 || opt.j < 0 || opt.i == opt.j) {
    std::cerr << "Invalid indices: --i and --j must be >=0 and different.\n";
    return 2;
  }
  std::vector<Row> rows;
  if (!load_keyframes(opt.csv, rows)) {
    std::cerr << "Failed to read keyframes CSV: " << opt.csv << "\n";
    return 2;
  }
  if (opt.i >= (int)rows.size() || opt.j >= (int)rows.size()) {
    std::cerr << "Index out of range. Keyframes in CSV: " << rows.size() << "\n";
    return 2;
  }
  std::map<std::string, GtPose> gt_of;
  if (!load_par(opt.par, gt_of)) {
    std::cerr << "Failed to read par file: " << opt.par << "\n";
    return 2;
  }
  const Row& ki = rows[(size_t)opt.i];
  const Row& kj = rows[(size_t)opt.j];
  const auto gi = gt_of.find(ki.image), gj = gt_of.find(kj.image);
  if (gi == gt_of.end() || gj == gt_of.end()) {
    std::cerr << "Image name not found in par file. Missing: " << (gi == gt_of.end() ? ki.image : "") << " "
              << (gj == gt_of.end() ? kj.image : "") << "\n";
    return 2;
  }
  const V3 gt_i = -(sfmx_host::transpose(gi->second.R) * gi->second.t);  // C = -R^T t
  const V3 gt_j = -(sfmx_host::transpose(gj->second.R) * gj->second.t);
  const V3 v_est = kj.centre - ki.centre, v_gt = gt_j - gt_i;
  const Mat3 R = rotation_between(v_est, v_gt);
  const double len_est = norm(v_est), len_gt = norm(v_gt);
  double s = 1.0;
  if (opt.with_scale && len_est > 1e-12) s = len_gt / len_est;
  const V3 t = gt_i - times(R * ki.centre, s);
  const V3 err_i = (times(R * ki.centre, s) + t) - gt_i;
  const V3 err_j = (times(R * kj.centre, s) + t) - gt_j;
  const double rmse = std::sqrt(0.5 * (dot(err_i, err_i) + dot(err_j, err_j)));

  std::cout.setf(std::ios::scientific);
  std::cout.precision(12);
  std::cout << "ATE (two keyframes)\n"
            << "  mode: " << (opt.with_scale ? "Sim(3)" : "SE(3)") << "\n"
            << "  keyframes: [" << opt.i << "] " << ki.image << "  ->  [" << opt.j << "] " << kj.image << "\n"
            << "  baseline_len_est: " << len_est << "\n"
            << "  baseline_len_gt : " << len_gt << "\n";
  if (opt.with_scale) std::cout << "  scale (s): " << s << "\n";
  std::cout << "  ATE_RMSE: " << rmse << "\n"
            << "  per_frame_error:\n"
            << "    " << ki.image << ": " << norm(err_i) << "\n"
            << "    " << kj.image << ": " << norm(err_j) << "\n";
  return 0;
}
