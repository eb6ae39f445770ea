// mesh.cpp — the optional sparse-mesh export of the CLI (--export-geometry mesh|both; reference T:1226-1461,1884-1906).
//
// Serial host code outside the hot path, here to complete the drop-in surface: the map points seen by one keyframe
// are projected into its image, thinned on a pixel grid in a shuffled order, triangulated in 2D (incremental
// Bowyer-Watson) and written as an ASCII PLY.  The result depends on three library behaviours of the reference
// build, all of them reproduced by using the same libstdc++ facilities with the same parameters: std::shuffle driven by
// std::mt19937(42), the iteration order of the map (unordered_map<int,.>) and the iteration order of the per-point
// edge table (unordered_map with the reference's edge hash and reserve()).  tests/test_host_math.py pins the output
// against the real reference functions (oracle/ref_harness.cpp) on committed fixtures.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <fstream>
#include <random>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>

#include "pipeline.hpp"

namespace sfmx_host {
namespace {

double turn(const V2& a, const V2& b, const V2& c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }  // T:1246-1249

// is p strictly inside the circumcircle of (a,b,c)?  Determinant test; its sign follows the triangle's orientation (T:1251-1270)
bool inside_circumcircle(const V2& a, const V2& b, const V2& c, const V2& p) {
  const double ax = a.x - p.x, ay = a.y - p.y, bx = b.x - p.x, by = b.y - p.y, cx = c.x - p.x, cy = c.y - p.y;
  const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  const double det = a2 * (bx * cy - by * cx) - b2 * (ax * cy - ay * cx) + c2 * (ax * by - ay * bx);
  return turn(a, b, c) > 0.0 ? det > 1e-12 : det < -1e-12;
}

struct Edge {
  int lo, hi;
  bool operator==(const Edge& o) const { return lo == o.lo && hi == o.hi; }
};
struct EdgeHash {  // T:1277-1282: the bucket order of the edge table decides the order of the new triangles
  std::size_t operator()(const Edge& e) const {
    return (std::size_t)((std::uint64_t)(std::uint32_t)e.lo * 2654435761u) ^ (std::size_t)(std::uint32_t)e.hi;
  }
};
using Tri = std::array<int, 3>;

// incremental Bowyer-Watson (T:1284-1362): points are inserted in index order into a triangulation that starts as one
// enclosing triangle; the triangles whose circumcircle holds the new point are removed and the hole is fanned from it
std::vector<Tri> delaunay_2d(const std::vector<V2>& pts) {
  const int n = (int)pts.size();
  if (n < 3) return {};
  double x0 = pts[0].x, x1 = pts[0].x, y0 = pts[0].y, y1 = pts[0].y;
  for (const V2& p : pts) {
    x0 = std::min(x0, p.x); x1 = std::max(x1, p.x);
    y0 = std::min(y0, p.y); y1 = std::max(y1, p.y);
  }
  const double span = std::max(x1 - x0, y1 - y0), mx = 0.5 * (x0 + x1), my = 0.5 * (y0 + y1);
  std::vector<V2> v = pts;
  v.push_back(V2{mx - 20.0 * span, my - 2.0 * span});
  v.push_back(V2{mx, my + 20.0 * span});
  v.push_back(V2{mx + 20.0 * span, my - 2.0 * span});
  std::vector<Tri> tris;
  if (turn(v[(size_t)n], v[(size_t)n + 1], v[(size_t)n + 2]) > 0.0) tris.push_back({n, n + 1, n + 2});
  else tris.push_back({n, n + 2, n + 1});

  std::vector<int> hit;
  std::vector<char> alive;
  std::vector<Tri> kept;
  for (int pi = 0; pi < n; ++pi) {
    const V2& p = v[(size_t)pi];
    hit.clear();
    for (int t = 0; t < (int)tris.size(); ++t)
      if (inside_circumcircle(v[(size_t)tris[(size_t)t][0]], v[(size_t)tris[(size_t)t][1]], v[(size_t)tris[(size_t)t][2]], p)) hit.push_back(t);
    // how often each undirected edge occurs among the removed triangles; boundary edges of the hole occur once
    std::unordered_map<Edge, int, EdgeHash> uses;
    uses.reserve(hit.size() * 3);
    for (int t : hit) {
      const Tri& q = tris[(size_t)t];
      for (int e = 0; e < 3; ++e) {
        const int a = q[(size_t)e], b = q[(size_t)((e + 1) % 3)];
        uses[Edge{std::min(a, b), std::max(a, b)}] += 1;
      }
    }
    if (!hit.empty()) {
      alive.assign(tris.size(), 1);
      for (int t : hit) alive[(size_t)t] = 0;
      kept.clear();
      kept.reserve(tris.size());
      for (size_t t = 0; t < tris.size(); ++t)
        if (alive[t]) kept.push_back(tris[t]);
      tris.swap(kept);
    }
    for (const auto& kv : uses) {  // table order = reference's order of the new triangles
      if (kv.second != 1) continue;
      const int a = kv.first.lo, b = kv.first.hi;
      if (turn(v[(size_t)a], v[(size_t)b], p) > 0.0) tris.push_back({a, b, pi});
      else tris.push_back({b, a, pi});
    }
  }
  std::vector<Tri> out;
  out.reserve(tris.size());
  for (const Tri& t : tris)
    if (t[0] < n && t[1] < n && t[2] < n) out.push_back(t);
  return out;
}

// T:1364-1377
bool project(const Mat3& K, const Pose& pose, const V3& Xw, int w, int h, V2& uv) {
  Mat3 Rwc;
  V3 twc;
  inv_wc(pose, Rwc, twc);
  const V3 Xc = (Rwc * Xw) + twc;
  if (!(Xc.z > 1e-8)) return false;
  const V3 ph = K * V3{Xc.x / Xc.z, Xc.y / Xc.z, 1.0};
  uv = V2{ph.x, ph.y};
  if (uv.x < 0.0 || uv.y < 0.0 || uv.x >= (double)w || uv.y >= (double)h) return false;
  return std::isfinite(uv.x) && std::isfinite(uv.y);
}

struct Cell {
  int cx, cy;
  bool operator==(const Cell& o) const { return cx == o.cx && cy == o.cy; }
};
struct CellHash {  // T:1416-1418
  std::size_t operator()(const Cell& k) const {
    return ((std::size_t)(std::uint32_t)k.cx * 73856093u) ^ ((std::size_t)(std::uint32_t)k.cy * 19349663u);
  }
};
}  // namespace

// T:1384-1461.  Empty outputs mean "skipped" (fewer than 50 usable points).
void build_sparse_mesh(const Mat3& K, const Pose& kf_pose, const MapState& map, int img_w, int img_h, int max_points, int grid_px,
                       double max_edge_px, std::vector<V3>& vertices, std::vector<std::array<int, 3>>& faces) {
  struct Sample { V2 uv; V3 Xw; };
  std::vector<Sample> seen;
  seen.reserve(map.pts.size());
  for (const auto& kv : map.pts) {  // map iteration order feeds the shuffle
    V2 uv;
    if (project(K, kf_pose, kv.second.Xw, img_w, img_h, uv)) seen.push_back(Sample{uv, kv.second.Xw});
  }
  vertices.clear();
  faces.clear();
  if ((int)seen.size() < 50) return;
  std::mt19937 rng(42);
  std::shuffle(seen.begin(), seen.end(), rng);
  // at most one sample per grid_px x grid_px pixel cell, first come first served
  const int cell = std::max(1, grid_px);
  std::unordered_set<Cell, CellHash> taken;
  taken.reserve((size_t)max_points * 2);
  std::vector<V2> uv_sel;
  vertices.reserve((size_t)max_points);
  uv_sel.reserve((size_t)max_points);
  for (const Sample& s : seen) {
    const Cell c{(int)std::floor(s.uv.x / (double)cell), (int)std::floor(s.uv.y / (double)cell)};
    if (!taken.insert(c).second) continue;
    uv_sel.push_back(s.uv);
    vertices.push_back(s.Xw);
    if ((int)vertices.size() >= max_points) break;
  }
  if ((int)vertices.size() < 50) return;
  for (const Tri& t : delaunay_2d(uv_sel)) {  // drop triangles with a long edge in the image
    const V2 &a = uv_sel[(size_t)t[0]], &b = uv_sel[(size_t)t[1]], &c = uv_sel[(size_t)t[2]];
    const double longest = std::max(std::hypot(a.x - b.x, a.y - b.y), std::max(std::hypot(b.x - c.x, b.y - c.y), std::hypot(c.x - a.x, c.y - a.y)));
    if (longest > max_edge_px) continue;
    faces.push_back(t);
  }
}

// T:1226-1244
void write_mesh_ply(const std::string& path, const std::vector<V3>& vertices, const std::vector<std::array<int, 3>>& faces) {
  std::ofstream f(path);
  if (!f) throw std::runtime_error("Failed to write: " + path);
  f << "ply\nformat ascii 1.0\n"
    << "element vertex " << vertices.size() << "\n"
    << "property float x\nproperty float y\nproperty float z\n"
    << "element face " << faces.size() << "\n"
    << "property list uchar int vertex_indices\n"
    << "end_header\n";
  for (const V3& p : vertices) f << p.x << " " << p.y << " " << p.z << "\n";
  for (const auto& t : faces) f << "3 " << t[0] << " " << t[1] << " " << t[2] << "\n";
}

}  // namespace sfmx_host

// test hook (CPU suite): points are added to a MapState in array order, as the pipeline does (pid = index)
extern "C" int sfmx_host_sparse_mesh(const double* K9, const double* pose12, const double* X, int n_pts, int w, int h, int max_points,
                                     int grid_px, double max_edge_px, double* verts_out, int verts_cap, int* faces_out, int faces_cap,
                                     int* n_faces) {
  using namespace sfmx_host;
  Mat3 K;
  std::copy(K9, K9 + 9, K.a);
  Pose pose;
  std::copy(pose12, pose12 + 9, pose.R.a);
  pose.t = {pose12[9], pose12[10], pose12[11]};
  Arena arena;
  MapState map(&arena);
  for (int p = 0; p < n_pts; p++) map.add(p, V3{X[3 * p], X[3 * p + 1], X[3 * p + 2]});
  std::vector<V3> v;
  std::vector<std::array<int, 3>> f;
  build_sparse_mesh(K, pose, map, w, h, max_points, grid_px, max_edge_px, v, f);
  for (int i = 0; i < (int)v.size() && i < verts_cap; i++) { verts_out[3 * i] = v[(size_t)i].x; verts_out[3 * i + 1] = v[(size_t)i].y; verts_out[3 * i + 2] = v[(size_t)i].z; }
  for (int i = 0; i < (int)f.size() && i < faces_cap; i++) { faces_out[3 * i] = f[(size_t)i][0]; faces_out[3 * i + 1] = f[(size_t)i][1]; faces_out[3 * i + 2] = f[(size_t)i][2]; }
  *n_faces = (int)f.size();
  return (int)v.size();
}
