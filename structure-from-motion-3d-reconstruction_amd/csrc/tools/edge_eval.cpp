// gt_keyframe_edge — ground-truth relative pose of two keyframes, and the error of an estimated pose-graph edge against
// it (reference cpp/tools/gt_keyframe_edge.cpp).  x_j = R_ij x_i + t_ij with R_ij = R_j R_i^T, t_ij = t_j - R_ij t_i from
// the Middlebury extrinsics; reported as Rodrigues vector + unit translation direction; with --edges the rotation error
// |log(R_est R_ij^T)| and the (sign-free) angle between the translation directions, in degrees.
// Same command line, output text and exit codes as the reference tool, quirks included: the edge file must have a
// `kind` column, which the pipeline's own posegraph_edges.csv (`...,inliers,is_loop`, T:1199-1209) does not -- feeding
// that file answers "Failed to read edges CSV" there and here.  tests/test_tools.py compares with the real tool.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <numbers>
#include <optional>
#include <sstream>
#include <string>
#include <vector>

#include "../host/host_math.hpp"

namespace {
using sfmx_host::Mat3;
using sfmx_host::V3;

// surrounding blanks, then at most one leading and one trailing quote of each kind (" first, then ')
std::string cleaned(std::string s) {
  auto blank = [](char c) { return std::isspace(static_cast<unsigned char>(c)) != 0; };
  size_t a = 0, b = s.size();
  while (a < b && blank(s[a])) ++a;
  while (b > a && blank(s[b - 1])) --b;
  for (char q : {'"', '\''}) {
    if (a < b && s[a] == q) ++a;
    if (a < b && s[b - 1] == q) --b;
  }
  return s.substr(a, b - a);
}
std::vector<std::string> comma_fields(const std::string& line) {  // no quoted commas
  std::vector<std::string> f;
  size_t start = 0;
  for (size_t i = 0; i <= line.size(); ++i)
    if (i == line.size() || line[i] == ',') {
      f.push_back(cleaned(line.substr(start, i - start)));
      start = i + 1;
    }
  return f;
}
template <class T>
std::optional<T> parsed(const std::string& raw) {  // the whole token must be consumed
  std::stringstream ss{cleaned(raw)};
  T v{};
  ss >> v;
  if (!ss.fail() && ss.eof()) return v;
  return std::nullopt;
}
std::vector<std::string> blank_fields(const std::string& line) {
  std::vector<std::string> f;
  std::istringstream ss(line);
  std::string tok;
  while (ss >> tok) f.push_back(tok);
  return f;
}

struct Camera { Mat3 R; V3 t; };
struct Kf { int id = 0; std::string image; };
struct Edge { int i = 0, j = 0; std::string kind; V3 rvec, t; };

// every line with at least 22 blank-separated tokens whose 21 numbers parse is a camera; the first one of a name wins
bool load_par(const std::string& path, std::map<std::string, Camera>& cams) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> tok = blank_fields(line);
    if (tok.size() < 22) continue;
    double v[21];
    bool ok = true;
    for (int k = 0; k < 21 && ok; k++) {
      const auto d = parsed<double>(tok[(size_t)k + 1]);
      ok = d.has_value();
      if (ok) v[k] = *d;
    }
    if (!ok) continue;
    Camera c;
    for (int k = 0; k < 9; k++) c.R.a[k] = v[9 + k];
    c.t = {v[18], v[19], v[20]};
    cams.emplace(tok[0], c);
  }
  return true;
}

// rows become indexable by kf_id: kept as read when the ids are 0,1,2,..., otherwise scattered into a table of the
// same length (ids outside it are dropped, holes stay empty)
bool load_keyframes(const std::string& path, std::vector<Kf>& out) {
  std::ifstream in(path);
  std::string line;
  if (!in || !std::getline(in, line)) return false;
  const std::vector<std::string> head = comma_fields(line);
  int c_id = -1, c_img = -1;
  for (int k = 0; k < (int)head.size(); ++k) {  // last match wins
    if (head[(size_t)k] == "kf_id") c_id = k;
    if (head[(size_t)k] == "image") c_img = k;
  }
  if (c_id < 0 || c_img < 0) return false;
  std::vector<Kf> rows;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> f = comma_fields(line);
    if ((int)f.size() <= std::max(c_id, c_img)) continue;
    const auto id = parsed<int>(f[(size_t)c_id]);
    if (!id) continue;
    rows.push_back(Kf{*id, f[(size_t)c_img]});
  }
  bool in_order = true;
  for (size_t k = 0; k < rows.size() && in_order; ++k) in_order = rows[k].id == (int)k;
  if (in_order) { out = std::move(rows); return true; }
  out.assign(rows.size(), Kf{});
  for (const Kf& r : rows)
    if (r.id >= 0 && (size_t)r.id < out.size()) out[(size_t)r.id] = r;
  return true;
}

bool load_edges(const std::string& path, std::vector<Edge>& out) {
  std::ifstream in(path);
  std::string line;
  if (!in || !std::getline(in, line)) return false;
  const std::vector<std::string> head = comma_fields(line);
  const char* names[9] = {"i", "j", "kind", "rvec_x", "rvec_y", "rvec_z", "t_x", "t_y", "t_z"};
  int col[9];
  int need = -1;
  for (int n = 0; n < 9; n++) {
    col[n] = -1;
    for (int k = 0; k < (int)head.size() && col[n] < 0; ++k)  // first match wins
      if (head[(size_t)k] == names[n]) col[n] = k;
    if (col[n] < 0) return false;
    need = std::max(need, col[n]);
  }
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const std::vector<std::string> f = comma_fields(line);
    if ((int)f.size() <= need) continue;
    const auto I = parsed<int>(f[(size_t)col[0]]), J = parsed<int>(f[(size_t)col[1]]);
    std::optional<double> d[6];
    bool ok = I.has_value() && J.has_value();
    for (int n = 0; n < 6; n++) {
      d[n] = parsed<double>(f[(size_t)col[3 + n]]);
      ok = ok && d[n].has_value();
    }
    if (!ok) continue;
    out.push_back(Edge{*I, *J, f[(size_t)col[2]], V3{*d[0], *d[1], *d[2]}, V3{*d[3], *d[4], *d[5]}});
  }
  return true;
}

// "--key value" with the key anywhere from argv[1] on and a value behind it; the first such key counts
std::optional<std::string> option(int argc, char** argv, const std::string& key) {
  for (int k = 1; k + 1 < argc; ++k)
    if (key == argv[k]) return std::string(argv[k + 1]);
  return std::nullopt;
}
V3 direction(const V3& v) {
  const double n = sfmx_host::norm(v);
  if (n < 1e-12) return {0, 0, 0};
  return {v.x / n, v.y / n, v.z / n};
}
double degrees(double r) { return r * (180.0 / std::numbers::pi); }
double clamped(double x) { return std::max(-1.0, std::min(1.0, x)); }
}  // namespace

int main(int argc, char** argv) {
  const auto par = option(argc, argv, "--par"), kfp = option(argc, argv, "--keyframes"), edp = option(argc, argv, "--edges");
  std::optional<int> ii, jj;
  if (const auto v = option(argc, argv, "--i")) ii = parsed<int>(*v);
  if (const auto v = option(argc, argv, "--j")) jj = parsed<int>(*v);
  bool emit_csv = false;
  for (int k = 1; k < argc; ++k) emit_csv |= std::string(argv[k]) == "--emit-csv";
  if (!par || !kfp || !ii || !jj) {
    std::cerr << "Usage:\n"
              << "  gt_keyframe_edge --par <*_par.txt> --keyframes <keyframes_camera_centers.csv> --i <kf_id> --j <kf_id> [--edges <posegraph_edges.csv>] [--emit-csv]\n\n"
              << "Outputs:\n"
              << "  - Ground-truth relative pose edge (Rodrigues rvec + translation direction) between the two keyframes.\n"
              << "  - If --edges is provided, also prints rotation and translation-direction errors versus the estimated edge.\n";
    return 2;
  }
  std::map<std::string, Camera> cams;
  if (!load_par(*par, cams)) {
    std::cerr << "Failed to read par file: " << *par << "\n";
    return 2;
  }
  std::vector<Kf> kfs;
  if (!load_keyframes(*kfp, kfs)) {
    std::cerr << "Failed to read keyframes CSV: " << *kfp << "\n";
    return 2;
  }
  if (*ii < 0 || *jj < 0 || (size_t)*ii >= kfs.size() || (size_t)*jj >= kfs.size()) {
    std::cerr << "Keyframe id out of range. Have " << kfs.size() << " keyframes.\n";
    return 2;
  }
  const std::string& img_i = kfs[(size_t)*ii].image;
  const std::string& img_j = kfs[(size_t)*jj].image;
  const auto ci = cams.find(img_i), cj = cams.find(img_j);
  if (ci == cams.end() || cj == cams.end()) {
    std::cerr << "Image not found in par file: " << (ci == cams.end() ? img_i : img_j) << "\n";
    return 2;
  }
  const Mat3 R_ij = cj->second.R * sfmx_host::transpose(ci->second.R);
  const V3 t_ij = cj->second.t - (R_ij * ci->second.t);
  const V3 rvec_gt = sfmx_host::so3_log(R_ij);
  const V3 tdir_gt = direction(t_ij);

  if (emit_csv) {  // one row in the column order the tool expects of an edge file
    std::cout << "i,j,kind,rvec_x,rvec_y,rvec_z,t_x,t_y,t_z\n";
    std::cout << *ii << "," << *jj << ",gt," << std::setprecision(10) << rvec_gt.x << "," << rvec_gt.y << "," << rvec_gt.z << ","
              << tdir_gt.x << "," << tdir_gt.y << "," << tdir_gt.z << "\n";
    return 0;
  }
  std::cout << std::fixed << std::setprecision(6);
  std::cout << "Keyframe edge (ground truth)\n"
            << "  i=" << *ii << " (" << img_i << ")\n"
            << "  j=" << *jj << " (" << img_j << ")\n"
            << "  rvec_gt = [" << rvec_gt.x << ", " << rvec_gt.y << ", " << rvec_gt.z << "]\n"
            << "  tdir_gt = [" << tdir_gt.x << ", " << tdir_gt.y << ", " << tdir_gt.z << "]\n";
  if (!edp) return 0;

  std::vector<Edge> edges;
  if (!load_edges(*edp, edges)) {
    std::cerr << "Failed to read edges CSV: " << *edp << "\n";
    return 2;
  }
  const Edge* e = nullptr;
  for (const Edge& c : edges)
    if (c.i == *ii && c.j == *jj) { e = &c; break; }
  if (!e) {
    std::cerr << "Edge (i,j)=(" << *ii << "," << *jj << ") not found in " << *edp << "\n";
    return 2;
  }
  const V3 tdir_est = direction(e->t);
  const Mat3 R_err = sfmx_host::so3_exp(e->rvec) * sfmx_host::transpose(R_ij);
  const double rot_err = degrees(sfmx_host::norm(sfmx_host::so3_log(R_err)));
  const double d1 = clamped(sfmx_host::dot(tdir_est, tdir_gt));
  const double d2 = clamped(sfmx_host::dot(tdir_est, V3{-tdir_gt.x, -tdir_gt.y, -tdir_gt.z}));
  const double tr_err = degrees(std::min(std::acos(d1), std::acos(d2)));
  std::cout << "\nEstimated edge (from posegraph_edges.csv)\n"
            << "  kind     = " << e->kind << "\n"
            << "  rvec_est = [" << e->rvec.x << ", " << e->rvec.y << ", " << e->rvec.z << "]\n"
            << "  tdir_est = [" << tdir_est.x << ", " << tdir_est.y << ", " << tdir_est.z << "]\n"
            << "\nErrors vs ground truth\n"
            << "  rotation error (deg)            = " << rot_err << "\n"
            << "  translation direction error (deg)= " << tr_err << "\n";
  return 0;
}
