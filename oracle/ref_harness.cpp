// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the *unmodified* reference translation unit where it lies under
// /root/reference and exposes its file-static hot-path functions through a
// flat C ABI so that tests can (a) validate the CPU restatement in
// oracle/sfm_oracle.cpp and (b) generate the golden vectors in tests/golden/.
//
// Nothing from the reference is copied into this repository: the TU and its
// headers are #included by absolute path.  Built only where /root/reference
// exists (this container); outputs go to oracle/_ref/ (git-ignored).
//
// Build quirk (documented in DESIGN.md §oracle): the reference's
// cpp/include/minijson.hpp:20 declares std::unordered_map<std::string, Value>
// inside Value, which libstdc++ 11 rejects (incomplete mapped type).  The JSON
// config reader is not on the hot path; we pre-include that one header with
// `unordered_map` spelled `map` (std::map accepts the incomplete type), then
// restore the token before the pipeline TU is included, so every container on
// the hot path (Keyframe::obs, MapState, track_hist, ...) is the real
// std::unordered_map.  `#pragma once` in minijson.hpp makes the TU's own
// #include of it a no-op.
#include <algorithm>
#include <array>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <numeric>
#include <optional>
#include <ostream>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#define unordered_map map
#include "/root/reference/cpp/include/minijson.hpp"
#undef unordered_map

// KLTTracker::lk_step / track_one are private (T:400-460); the harness needs
// them individually.  All std headers are already included above, so this
// only touches the reference's own classes.
#define private public
#define main sfm_reference_main
#include "/root/reference/cpp/src/templering_sfm.cpp"
#undef main
#undef private

namespace {

GrayImage wrap_image(const std::uint8_t* pix, int w, int h) {
  GrayImage im;
  im.w = w;
  im.h = h;
  im.pix.assign(pix, pix + (size_t)w * (size_t)h);
  return im;
}

Mat33 wrap_m33(const double* a) {
  Mat33 m{};
  for (int i = 0; i < 9; i++) m.a[i] = a[i];
  return m;
}

LKConfig make_lk(int max_tracks, int min_tracks, double quality, int min_distance,
                 int levels, int radius, int iters, double fb) {
  LKConfig c;
  c.max_tracks = max_tracks;
  c.min_tracks = min_tracks;
  c.quality = quality;
  c.min_distance = min_distance;
  c.pyr_levels = levels;
  c.win_radius = radius;
  c.iters = iters;
  c.fb_thresh = fb;
  return c;
}

struct TrackerBox {
  KLTTracker trk;
  explicit TrackerBox(LKConfig c) : trk(c) {}
};

}  // namespace

extern "C" {

int ref_main(int argc, char** argv) { return sfm_reference_main(argc, argv); }

// T:200-218
void ref_downsample2(const std::uint8_t* pix, int w, int h, std::uint8_t* out) {
  GrayImage o = downsample2(wrap_image(pix, w, h));
  std::memcpy(out, o.pix.data(), o.pix.size());
}

// T:183-198
double ref_sample_bilinear(const std::uint8_t* pix, int w, int h, double x, double y) {
  return sample_bilinear(wrap_image(pix, w, h), x, y);
}

// T:237-302; returns number of corners written (capacity = max_corners)
int ref_shi_tomasi(const std::uint8_t* pix, int w, int h, int max_corners, double quality,
                   int min_dist, double* out_xy) {
  auto pts = shi_tomasi(wrap_image(pix, w, h), max_corners, quality, min_dist);
  for (size_t i = 0; i < pts.size(); i++) {
    out_xy[2 * i] = pts[i].x;
    out_xy[2 * i + 1] = pts[i].y;
  }
  return (int)pts.size();
}

// T:424-460 on single-level images
void ref_lk_step(const std::uint8_t* i0, const std::uint8_t* i1, int w, int h, int radius,
                 double x, double y, double* out2) {
  KLTTracker t(make_lk(1, 1, 0.01, 8, 1, radius, 1, 1.0));
  Vec2 s = t.lk_step(wrap_image(i0, w, h), wrap_image(i1, w, h), Vec2{x, y});
  out2[0] = s.x;
  out2[1] = s.y;
}

// T:356-362 for n points: forward a->b, backward b->a from the forward result.
void ref_klt_track(const std::uint8_t* ia, const std::uint8_t* ib, int w, int h, int levels,
                   int radius, int iters, int n, const double* xy_in, double* xy_fwd,
                   double* xy_back) {
  KLTTracker t(make_lk(1, 1, 0.01, 8, levels, radius, iters, 1.0));
  Pyramid pa = build_pyr(wrap_image(ia, w, h), levels);
  Pyramid pb = build_pyr(wrap_image(ib, w, h), levels);
  for (int i = 0; i < n; i++) {
    Vec2 p0{xy_in[2 * i], xy_in[2 * i + 1]};
    Vec2 p1 = t.track_one_public(pa, pb, p0);
    Vec2 pbk = t.track_one_public(pb, pa, p1);
    xy_fwd[2 * i] = p1.x;
    xy_fwd[2 * i + 1] = p1.y;
    xy_back[2 * i] = pbk.x;
    xy_back[2 * i + 1] = pbk.y;
  }
}

// Stateful tracker (T:323-466)
void* ref_tracker_create(int max_tracks, int min_tracks, double quality, int min_distance,
                         int levels, int radius, int iters, double fb) {
  return new TrackerBox(make_lk(max_tracks, min_tracks, quality, min_distance, levels, radius,
                                iters, fb));
}
void ref_tracker_destroy(void* h) { delete static_cast<TrackerBox*>(h); }

// step: returns survivor count; arrays must hold max_tracks entries
int ref_tracker_step(void* h, const std::uint8_t* pix, int w, int hgt, double* prev_xy,
                     double* cur_xy, int* ids) {
  auto& t = static_cast<TrackerBox*>(h)->trk;
  auto out = t.step(wrap_image(pix, w, hgt));
  for (size_t i = 0; i < out.ids.size(); i++) {
    prev_xy[2 * i] = out.prev_pts[i].x;
    prev_xy[2 * i + 1] = out.prev_pts[i].y;
    cur_xy[2 * i] = out.cur_pts[i].x;
    cur_xy[2 * i + 1] = out.cur_pts[i].y;
    ids[i] = out.ids[i];
  }
  return (int)out.ids.size();
}
int ref_tracker_tracks(void* h, double* xy, int* ids) {
  auto& t = static_cast<TrackerBox*>(h)->trk;
  const auto& tr = t.tracks();
  for (size_t i = 0; i < tr.size(); i++) {
    xy[2 * i] = tr[i].p.x;
    xy[2 * i + 1] = tr[i].p.y;
    ids[i] = tr[i].id;
  }
  return (int)tr.size();
}

// libstdc++ draw sequence used at T:657-665
void ref_uniform_draws(unsigned seed, int n, int count, int* out) {
  std::mt19937 rng(seed);
  std::uniform_int_distribution<int> uni(0, n - 1);
  for (int i = 0; i < count; i++) out[i] = uni(rng);
}

// T:471-501
int ref_normalize_points(const double* K9, const double* px, int n, double* out) {
  try {
    const Mat33 Kinv = invert_K(wrap_m33(K9));
    for (int i = 0; i < n; i++) {
      Vec2 q = norm_point(Kinv, Vec2{px[2 * i], px[2 * i + 1]});
      out[2 * i] = q.x;
      out[2 * i + 1] = q.y;
    }
    return 0;
  } catch (...) {
    return 1;
  }
}

// cpp/include/linalg.hpp:133-201
void ref_jacobi_eig_sym(const double* A, int N, int iters, double* w, double* V) {
  std::vector<double> a(A, A + (size_t)N * N);
  auto e = sfm::jacobi_eig_sym(a, N, iters);
  for (int i = 0; i < N; i++) w[i] = e.w[i];
  for (int i = 0; i < N * N; i++) V[i] = e.V[i];
}

// T:537-593
void ref_svd3(const double* A9, double* U9, double* s3, double* V9) {
  auto r = svd3(wrap_m33(A9));
  for (int i = 0; i < 9; i++) {
    U9[i] = r.U.a[i];
    V9[i] = r.V.a[i];
  }
  for (int i = 0; i < 3; i++) s3[i] = r.s[i];
}

// T:609-627 (xn,yn already normalised)
void ref_eight_point_E(const double* xn, const double* yn, int n, const int* idx8, double* E9) {
  std::vector<Vec2> a((size_t)n), b((size_t)n);
  for (int i = 0; i < n; i++) {
    a[i] = {xn[2 * i], xn[2 * i + 1]};
    b[i] = {yn[2 * i], yn[2 * i + 1]};
  }
  std::vector<int> id(idx8, idx8 + 8);
  Mat33 E = eight_point_E(a, b, id);
  for (int i = 0; i < 9; i++) E9[i] = E.a[i];
}

// T:629-638
double ref_sampson_err(const double* E9, double x, double y, double xp, double yp) {
  return sampson_err(wrap_m33(E9), Vec2{x, y}, Vec2{xp, yp});
}

// T:646-761.  returns 1 if a pose was found.  inliers must hold n ints.
int ref_find_E_ransac(const double* K9, const double* pi, const double* pj, int n, int iters,
                      double thr, int min_inliers, double* R9, double* t3, int* inliers,
                      int* n_inl) {
  std::vector<Vec2> a((size_t)n), b((size_t)n);
  for (int i = 0; i < n; i++) {
    a[i] = {pi[2 * i], pi[2 * i + 1]};
    b[i] = {pj[2 * i], pj[2 * i + 1]};
  }
  auto r = find_E_ransac(wrap_m33(K9), a, b, iters, thr, min_inliers);
  if (!r) {
    *n_inl = 0;
    return 0;
  }
  for (int i = 0; i < 9; i++) R9[i] = r->R_ji.a[i];
  t3[0] = r->t_ji.x;
  t3[1] = r->t_ji.y;
  t3[2] = r->t_ji.z;
  *n_inl = (int)r->inliers.size();
  for (size_t i = 0; i < r->inliers.size(); i++) inliers[i] = r->inliers[i];
  return 1;
}

// T:1477-1516.  Poses are camera->world (R row-major, t = centre).
void ref_triangulate_dlt(const double* K9, const double* Ri, const double* ti, const double* Rj,
                         const double* tj, const double* ui, const double* uj, double* X3) {
  PoseCW a{wrap_m33(Ri), Vec3{ti[0], ti[1], ti[2]}};
  PoseCW b{wrap_m33(Rj), Vec3{tj[0], tj[1], tj[2]}};
  Vec3 X = triangulate_dlt(wrap_m33(K9), a, b, Vec2{ui[0], ui[1]}, Vec2{uj[0], uj[1]});
  X3[0] = X.x;
  X3[1] = X.y;
  X3[2] = X.z;
}

// cpp/include/dense.hpp:54-93.  returns 0 ok, 1 if the reference threw.
int ref_solve_gauss(const double* A, const double* b, int n, double* x) {
  sfm::DMat M(n, n, 0.0);
  sfm::DVec v(n, 0.0);
  for (int i = 0; i < n * n; i++) M.a[(size_t)i] = A[i];
  for (int i = 0; i < n; i++) v[i] = b[i];
  try {
    sfm::DVec r = sfm::solve_gauss(M, v);
    for (int i = 0; i < n; i++) x[i] = r[i];
    return 0;
  } catch (...) {
    return 1;
  }
}

int ref_inv3(const double* A9, double* inv9) { return sfm::inv3(A9, inv9) ? 1 : 0; }

void ref_so3_exp(const double* w3, double* R9) {
  Mat33 R = sfm::so3_exp(Vec3{w3[0], w3[1], w3[2]});
  for (int i = 0; i < 9; i++) R9[i] = R.a[i];
}
void ref_so3_log(const double* R9, double* w3) {
  Vec3 w = sfm::so3_log(wrap_m33(R9));
  w3[0] = w.x;
  w3[1] = w.y;
  w3[2] = w.z;
}

// T:848-1097.  Keyframes: n_kf poses (camera->world R 9 + t 3 each), kf_id = index.
// Map points are inserted with MapState::add in array order (pid = index, tid = index) and
// observations appended in the order given (obs_ptr CSR over points: kf id + uv).
// Poses are updated in place.
void ref_bundle_adjust_window(const double* K9, int n_kf, double* poses12, int n_pts,
                              const double* X, const int* obs_ptr, const int* obs_kf,
                              const double* obs_uv, int window, int iters, int max_points,
                              double huber, double lambda) {
  std::vector<Keyframe> kfs((size_t)n_kf);
  for (int k = 0; k < n_kf; k++) {
    kfs[k].kf_id = k;
    kfs[k].frame_idx = k;
    kfs[k].pose.R = wrap_m33(poses12 + 12 * k);
    kfs[k].pose.t = Vec3{poses12[12 * k + 9], poses12[12 * k + 10], poses12[12 * k + 11]};
  }
  MapState map;
  for (int p = 0; p < n_pts; p++) {
    map.add(p, Vec3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
    for (int o = obs_ptr[p]; o < obs_ptr[p + 1]; o++)
      map.add_obs(p, obs_kf[o], Vec2{obs_uv[2 * o], obs_uv[2 * o + 1]});
  }
  BAConfig cfg;
  cfg.window = window;
  cfg.iters = iters;
  cfg.max_points = max_points;
  cfg.huber_delta = huber;
  cfg.lambda = lambda;
  bundle_adjust_window(wrap_m33(K9), kfs, map, cfg);
  for (int k = 0; k < n_kf; k++) {
    for (int i = 0; i < 9; i++) poses12[12 * k + i] = kfs[k].pose.R.a[i];
    poses12[12 * k + 9] = kfs[k].pose.t.x;
    poses12[12 * k + 10] = kfs[k].pose.t.y;
    poses12[12 * k + 11] = kfs[k].pose.t.z;
  }
}

// Iteration order of MapState::pts after the same insert sequence (pids in visit order).
void ref_map_iteration_order(int n_pts, int* order) {
  MapState map;
  for (int p = 0; p < n_pts; p++) map.add(p, Vec3{0, 0, 0});
  int k = 0;
  for (auto& kv : map.pts) order[k++] = kv.first;
}

// T:1131-1197.  centres3 updated in place; returns 1 if solved.
int ref_posegraph_optimize_centers(int n_kf, const double* R9s, double* centres3, int n_edges,
                                   const int* ei, const int* ej, const double* eR9,
                                   const double* et3, const int* is_loop) {
  std::vector<Keyframe> kfs((size_t)n_kf);
  for (int k = 0; k < n_kf; k++) {
    kfs[k].kf_id = k;
    kfs[k].pose.R = wrap_m33(R9s + 9 * k);
    kfs[k].pose.t = Vec3{centres3[3 * k], centres3[3 * k + 1], centres3[3 * k + 2]};
  }
  std::vector<PGEdge> edges((size_t)n_edges);
  for (int e = 0; e < n_edges; e++) {
    edges[e].i = ei[e];
    edges[e].j = ej[e];
    edges[e].R_ji = wrap_m33(eR9 + 9 * e);
    edges[e].t_ji = Vec3{et3[3 * e], et3[3 * e + 1], et3[3 * e + 2]};
    edges[e].is_loop = is_loop[e] != 0;
  }
  bool ok = posegraph_optimize_centers(kfs, edges);
  for (int k = 0; k < n_kf; k++) {
    centres3[3 * k] = kfs[k].pose.t.x;
    centres3[3 * k + 1] = kfs[k].pose.t.y;
    centres3[3 * k + 2] = kfs[k].pose.t.z;
  }
  return ok ? 1 : 0;
}

// T:1100-1122
void ref_global_desc_32(const std::uint8_t* pix, int w, int h, float* out1024) {
  auto v = global_desc_32(wrap_image(pix, w, h));
  for (int i = 0; i < 1024; i++) out1024[i] = v[(size_t)i];
}

// T:1384-1461 (sparse mesh of the points seen by one keyframe).  Points are inserted with MapState::add in array order.
// Returns the number of vertices; *n_faces the number of faces.
int ref_sparse_mesh(const double* K9, const double* pose12, const double* X, int n_pts, int w, int h, int max_points,
                    int grid_px, double max_edge_px, double* verts_out, int verts_cap, int* faces_out, int faces_cap, int* n_faces) {
  Keyframe kf;
  kf.pose.R = wrap_m33(pose12);
  kf.pose.t = Vec3{pose12[9], pose12[10], pose12[11]};
  MapState map;
  for (int p = 0; p < n_pts; p++) map.add(p, Vec3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
  std::vector<Vec3> v;
  std::vector<std::array<int, 3>> f;
  build_mesh_from_sparse_points(wrap_m33(K9), kf, map.pts, w, h, max_points, grid_px, max_edge_px, v, f);
  for (int i = 0; i < (int)v.size() && i < verts_cap; i++) { verts_out[3 * i] = v[i].x; verts_out[3 * i + 1] = v[i].y; verts_out[3 * i + 2] = v[i].z; }
  for (int i = 0; i < (int)f.size() && i < faces_cap; i++) { faces_out[3 * i] = f[i][0]; faces_out[3 * i + 1] = f[i][1]; faces_out[3 * i + 2] = f[i][2]; }
  *n_faces = (int)f.size();
  return (int)v.size();
}

// cpp/include/pgm_io.hpp:36-54: 0 = ok (size and byte sum of the pixels), 1 = it threw (message in err)
int ref_read_pgm(const char* path, int* w, int* h, unsigned long long* checksum, char* err, int cap) {
  try {
    const sfm::GrayImage im = sfm::read_pgm(path);
    *w = im.w;
    *h = im.h;
    unsigned long long s = 0;
    for (size_t i = 0; i < im.pix.size(); i++) s += (unsigned long long)im.pix[i] * (i % 251 + 1);
    *checksum = s;
    return 0;
  } catch (const std::exception& e) {
    std::snprintf(err, (size_t)cap, "%s", e.what());
    return 1;
  }
}

// minijson::parse + jpick({"cpp",sec,key},{"common",sec,key}) + jint / jdouble / jstring (T:65-106).
// kind 0 = int, 1 = double, 2 = string.  1 = found, 0 = absent or of another type, -1 = the parser threw (message in text_out)
int ref_config_lookup(const char* json, const char* section, const char* key, int kind, double* num_out, char* text_out, int cap) {
  try {
    const minijson::Value root = minijson::parse(json);
    const minijson::Value* v = jpick(root, {"cpp", section, key}, {"common", section, key});
    if (kind == 0) { const auto r = jint(v); if (!r) return 0; *num_out = *r; return 1; }
    if (kind == 1) { const auto r = jdouble(v); if (!r) return 0; *num_out = *r; return 1; }
    const auto r = jstring(v);
    if (!r) return 0;
    std::snprintf(text_out, (size_t)cap, "%s", r->c_str());
    return 1;
  } catch (const std::exception& e) {
    std::snprintf(text_out, (size_t)cap, "%s", e.what());
    return -1;
  }
}

}  // extern "C"
