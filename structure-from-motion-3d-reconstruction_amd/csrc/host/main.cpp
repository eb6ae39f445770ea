// templering_sfm — drop-in CLI for the reference's cpp/ pipeline (T:1518-1917) running its hot path
// on one MI355X through libsfmx.  Same positional arguments, flags, config.json lookup, stdout lines,
// CSV / PLY outputs and exit codes (2 usage, 0 ok/help, 1 "ERROR: ...").
//
// --export-geometry mesh|both writes templeRing_mesh_sparse_kf<k>.ply through mesh.cpp (T:1884-1906).
#include <cstdlib>
#include <filesystem>
#include <iostream>
#include <optional>
#include <string>

#include "cli_io.hpp"
#include "pipeline.hpp"

namespace fs = std::filesystem;
using namespace sfmx_host;
using namespace sfmx_cli;

namespace {

enum class ExportGeometry { NONE, POINTCLOUD, MESH, BOTH };
std::optional<ExportGeometry> parse_export_geometry(const std::string& s) {  // T:42-51
  if (s == "none") return ExportGeometry::NONE;
  if (s == "pointcloud") return ExportGeometry::POINTCLOUD;
  if (s == "mesh") return ExportGeometry::MESH;
  if (s == "mesh_stereo") return ExportGeometry::MESH;
  if (s == "both") return ExportGeometry::BOTH;
  return std::nullopt;
}

std::string read_text_file(const fs::path& p) {
  std::ifstream f(p);
  if (!f) throw std::runtime_error("Failed to open: " + p.string());
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

// frames read lazily from <root>/templeRing_pgm/<stem>.pgm, exactly when the reference reads them
struct PgmFrames : FrameSource {
  fs::path dir;
  std::vector<std::string> names;
  int w = 0, h = 0;
  int count() const override { return (int)names.size(); }
  int width() const override { return w; }
  int height() const override { return h; }
  fs::path path_of(int fi) const { return dir / (fs::path(names[(size_t)fi]).replace_extension(".pgm")); }
  void load(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr) override {
    const Gray g = read_pgm(path_of(fi).string());
    if (g.w != w || g.h != h) throw std::runtime_error("Image size differs from the first frame: " + path_of(fi).string());
    const int rc = sfmx_pyramid_upload(ctx, pyr, g.pix.data());
    if (rc != SFMX_OK) throw SfmxFailure(rc, std::string("sfmx: pyramid_upload: ") + sfmx_last_error(ctx));
    sfmx_sync(ctx);  // g goes out of scope
  }
};

void echo_line(const std::string& s) { std::cout << s; }

}  // namespace

int main(int argc, char** argv) {
  // the pipeline keeps five streams busy; HIP's default of 4 hardware queues makes lanes share one
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  try {
    if (argc < 3) {
      std::cerr << "Usage: " << argv[0] << " <templering_root> <out_dir> [frames] [options]\n"
                << "Input must be PGM images (P5) in <templering_root>/templeRing_pgm/\n"
                << "and par/ang files in <templering_root>/templeRing/.\n\n"
                << "Options:\n"
                << "  --config <path>           Config JSON (defaults to ./config.json when present)\n"
                << "  --export-geometry <none|pointcloud|mesh|both>\n"
                << "      none: no .ply geometry outputs\n"
                << "      pointcloud: write templeRing_sparse_points.ply\n"
                << "      mesh: write templeRing_mesh_sparse_kf<k>.ply (2D Delaunay on projected sparse points)\n"
                << "      both: write both pointcloud and mesh\n"
                << "  --mesh-kf <k>            Keyframe index used for 2D projection (default 0)\n"
                << "  --mesh-max-points <n>    Max vertices in mesh (default 2500)\n"
                << "  --mesh-grid-px <px>      Pixel grid subsampling cell size (default 4)\n"
                << "  --mesh-max-edge-px <px>  Reject triangles with any edge longer than this (default 80)\n";
      return 2;
    }
    const fs::path root = fs::path(argv[1]);
    const fs::path out = fs::path(argv[2]);
    int frames = 12;
    bool frames_from_cli = false;
    fs::path config_path;
    bool have_config = false;
    int argi = 3;
    if (argc >= 4) {
      const std::string a3 = argv[3];
      if (!a3.empty() && a3[0] != '-') {
        frames = std::stoi(a3);
        frames_from_cli = true;
        argi = 4;
      }
    }
    ExportGeometry export_geom = ExportGeometry::POINTCLOUD;
    bool export_geom_from_cli = false;
    int mesh_kf = 0, mesh_max_points = 2500, mesh_grid_px = 4;
    double mesh_max_edge_px = 80.0;
    bool mesh_kf_cli = false, mesh_max_points_cli = false, mesh_grid_px_cli = false, mesh_max_edge_px_cli = false;
    PipelineConfig pc;
    while (argi < argc) {
      const std::string flag = argv[argi++];
      auto need = [&](const std::string& name) -> std::string {
        if (argi >= argc) throw std::runtime_error("Missing value for " + name);
        return std::string(argv[argi++]);
      };
      if (flag == "--config") { config_path = fs::path(need(flag)); have_config = true; }
      else if (flag == "--export-geometry") {
        const std::string v = need(flag);
        const auto eg = parse_export_geometry(v);
        if (!eg) throw std::runtime_error("Invalid --export-geometry value: " + v);
        export_geom = *eg;
        export_geom_from_cli = true;
      }
      else if (flag == "--mesh-kf") { mesh_kf = std::stoi(need(flag)); mesh_kf_cli = true; }
      else if (flag == "--mesh-max-points") { mesh_max_points = std::stoi(need(flag)); mesh_max_points_cli = true; }
      else if (flag == "--mesh-grid-px") { mesh_grid_px = std::stoi(need(flag)); mesh_grid_px_cli = true; }
      else if (flag == "--mesh-max-edge-px") { mesh_max_edge_px = std::stod(need(flag)); mesh_max_edge_px_cli = true; }
      else if (flag == "-h" || flag == "--help") { std::cerr << "Run without args to see usage.\n"; return 0; }
      else throw std::runtime_error("Unknown option: " + flag);
    }
    if (!have_config) {
      const fs::path local = fs::path("config.json");
      if (fs::exists(local)) { config_path = local; have_config = true; }
    }
    std::optional<Json> cfg;
    if (have_config) {
      try {
        cfg = JsonParser(read_text_file(config_path)).parse();
      } catch (const std::exception& e) {
        throw std::runtime_error("Failed to parse config.json: " + config_path.string() + " | " + e.what());
      }
    }
    if (cfg) {  // T:1631-1676
      if (!frames_from_cli)
        if (auto v = jint(jpick(*cfg, "system", "frames"))) frames = std::max(1, *v);
      if (!export_geom_from_cli)
        if (auto s = jstring(jpick(*cfg, "outputs", "export_geometry")))
          if (const auto eg = parse_export_geometry(*s)) export_geom = *eg;
      if (!mesh_kf_cli) if (auto v = jint(jpick(*cfg, "mesh_sparse", "kf"))) mesh_kf = *v;  // T:1642-1653
      if (!mesh_max_points_cli) if (auto v = jint(jpick(*cfg, "mesh_sparse", "max_points"))) mesh_max_points = *v;
      if (!mesh_grid_px_cli) if (auto v = jint(jpick(*cfg, "mesh_sparse", "grid_px"))) mesh_grid_px = *v;
      if (!mesh_max_edge_px_cli) if (auto v = jdouble(jpick(*cfg, "mesh_sparse", "max_edge_px"))) mesh_max_edge_px = *v;
      if (auto v = jint(jpick(*cfg, "klt", "max_tracks"))) pc.klt.max_tracks = *v;
      if (auto v = jint(jpick(*cfg, "klt", "min_tracks"))) pc.klt.min_tracks = *v;
      if (auto v = jdouble(jpick(*cfg, "klt", "quality"))) pc.klt.quality = *v;
      if (auto v = jint(jpick(*cfg, "klt", "min_distance"))) pc.klt.min_distance = *v;
      if (auto v = jint(jpick(*cfg, "klt", "pyr_levels"))) pc.klt.pyr_levels = *v;
      if (auto v = jint(jpick(*cfg, "klt", "win_radius"))) pc.klt.win_radius = *v;
      if (auto v = jint(jpick(*cfg, "klt", "iters"))) pc.klt.iters = *v;
      if (auto v = jdouble(jpick(*cfg, "klt", "fb_thresh"))) pc.klt.fb_thresh = *v;
      if (auto v = jint(jpick(*cfg, "keyframe", "min_gap"))) pc.kf_min_gap = *v;
      if (auto v = jint(jpick(*cfg, "keyframe", "min_inliers"))) pc.kf_min_inliers = *v;
      if (auto v = jdouble(jpick(*cfg, "keyframe", "parallax_px"))) pc.kf_parallax_px = *v;
      if (auto v = jint(jpick(*cfg, "ba", "window"))) pc.ba.window = *v;
      if (auto v = jint(jpick(*cfg, "ba", "iters"))) pc.ba.iters = *v;
      if (auto v = jint(jpick(*cfg, "ba", "max_points"))) pc.ba.max_points = *v;
      if (auto v = jdouble(jpick(*cfg, "ba", "huber_delta"))) pc.ba.huber_delta = *v;
      if (auto v = jdouble(jpick(*cfg, "ba", "lambda"))) pc.ba.lambda = *v;
    }
    pc.frames = frames;
    pc.export_pointcloud = (export_geom == ExportGeometry::POINTCLOUD || export_geom == ExportGeometry::BOTH);

    const fs::path par = root / "templeRing" / "templeR_par.txt";
    const fs::path ang = root / "templeRing" / "templeR_ang.txt";
    const auto recs = read_par(par.string());
    const auto angs = read_ang(ang.string());
    if (recs.empty()) throw std::runtime_error("No records in par file.");
    const Mat3 K = recs.front().K;

    PgmFrames src;
    src.dir = root / "templeRing_pgm";
    std::vector<FrameMeta> meta;
    for (const auto& r : recs) {
      src.names.push_back(r.img);
      FrameMeta m;
      m.name = r.img;
      const auto it = angs.find(r.img);
      if (it != angs.end()) { m.has_ang = true; m.lat = it->second.lat; m.lon = it->second.lon; }
      meta.push_back(m);
    }
    if (std::min(frames, (int)recs.size()) > 0) {  // image size from the first frame (the reference reads it first too)
      const Gray g0 = read_pgm(src.path_of(0).string());
      src.w = g0.w;
      src.h = g0.h;
    }

    sfmx_ctx* ctx = nullptr;
    const char* dev_env = std::getenv("SFMX_DEVICE");
    const int rc = sfmx_ctx_create(dev_env ? std::atoi(dev_env) : 0, &ctx);
    if (rc != SFMX_OK) throw std::runtime_error("no usable MI355X (gfx950) device: sfmx_ctx_create failed (there is no CPU fallback)");
    struct CtxGuard { sfmx_ctx* c; ~CtxGuard() { sfmx_ctx_destroy(c); } } guard{ctx};

    PipelineResult res;
    run_pipeline(ctx, src, meta, K, pc, res, echo_line);
    const size_t printed = res.log.size();
    write_outputs(out.string(), pc, meta, res);
    if (export_geom == ExportGeometry::MESH || export_geom == ExportGeometry::BOTH) {  // T:1884-1906
      if (res.kfs.empty()) {
        std::cerr << "WARN: mesh export skipped (no keyframes).\n";
      } else {
        const int kidx = std::max(0, std::min(mesh_kf, (int)res.kfs.size() - 1));
        const Keyframe& mkf = res.kfs[(size_t)kidx];
        const Gray im = read_pgm((src.dir / fs::path(mkf.img_name).replace_extension(".pgm")).string());  // its size bounds the projection
        std::vector<V3> verts;
        std::vector<std::array<int, 3>> faces;
        build_sparse_mesh(K, mkf.pose, res.map, im.w, im.h, mesh_max_points, mesh_grid_px, mesh_max_edge_px, verts, faces);
        if (verts.empty() || faces.empty()) std::cerr << "WARN: mesh export skipped (insufficient projected points or no valid triangles).\n";
        else write_mesh_ply((out / (std::string("templeRing_mesh_sparse_kf") + std::to_string(kidx) + ".ply")).string(), verts, faces);
      }
    }
    std::cout << res.log.substr(printed);
    if (std::getenv("SFMX_TIMING")) {
      const StageClock& c = res.clock;
      std::cerr << "[sfmx] total " << c.total << " s | klt " << c.klt << " | shi " << c.shi << " | ransac " << c.ransac << " | ba " << c.ba
                << " | upload " << c.upload << " | host-tri " << c.host << " | lk_steps " << c.lk_steps << "\n";
    }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "ERROR: " << e.what() << "\n";
    return 1;
  }
}
