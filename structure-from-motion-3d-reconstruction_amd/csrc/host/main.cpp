// templering_sfm — drop-in CLI for the reference's cpp/ pipeline (T:1518-1917) running its hot path
// on one MI355X through libsfmx.  Same positional arguments, flags, config.json lookup, stdout lines,
// CSV / PLY outputs and exit codes (2 usage, 0 ok/help, 1 "ERROR: ...").
//
// --export-geometry mesh|both writes templeRing_mesh_sparse_kf<k>.ply through mesh.cpp (T:1884-1906).
#include <chrono>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <optional>
#include <string>
#include <thread>

#include "cli_io.hpp"
#include "pipeline.hpp"

namespace fs = std::filesystem;
using namespace sfmx_host;
using namespace sfmx_cli;

namespace {

// ---- settings of one run and where each of them may come from ---------------------------------------------------
// Precedence of the reference (T:1537-1676): command line > config "cpp" section > config "common" section > built-in
// default.  Every setting is one row of a table: its command-line flag (if it has one), its config.json location (if it
// has one) and a typed slot; parsing and the config overlay are loops over the tables.
enum class Geometry { None, PointCloud, Mesh, Both };

bool geometry_from_name(const std::string& name, Geometry& out) {  // accepted spellings: T:42-51
  static const std::pair<const char*, Geometry> names[] = {{"none", Geometry::None}, {"pointcloud", Geometry::PointCloud}, {"mesh", Geometry::Mesh},
                                                            {"mesh_stereo", Geometry::Mesh}, {"both", Geometry::Both}};
  for (const auto& kv : names)
    if (name == kv.first) { out = kv.second; return true; }
  return false;
}

struct Settings {
  int frames = 12;
  Geometry geometry = Geometry::PointCloud;
  int mesh_kf = 0, mesh_max_points = 2500, mesh_grid_px = 4;
  double mesh_max_edge_px = 80.0;
  PipelineConfig pipe;
  std::string config_path;
  bool config_given = false;
};

// a typed destination inside Settings
struct Slot {
  enum Type { Int, Real, GeometryName, Path } type;
  void* at;
  void from_text(const std::string& text, const std::string& flag) const {  // command-line value
    switch (type) {
      case Int: *static_cast<int*>(at) = std::stoi(text); break;
      case Real: *static_cast<double*>(at) = std::stod(text); break;
      case Path: *static_cast<std::string*>(at) = text; break;
      case GeometryName:
        if (!geometry_from_name(text, *static_cast<Geometry*>(at))) throw std::runtime_error("Invalid " + flag + " value: " + text);
        break;
    }
  }
  void from_config(Json::Ref v) const {  // a value of the wrong JSON type is ignored, as in the reference's getters (T:84-106)
    switch (type) {
      case Int: if (const auto i = as_int(v)) *static_cast<int*>(at) = *i; break;
      case Real: if (const auto d = as_number(v)) *static_cast<double*>(at) = *d; break;
      case GeometryName: if (const auto t = as_string(v)) (void)geometry_from_name(*t, *static_cast<Geometry*>(at)); break;
      case Path: break;
    }
  }
};
struct Row {
  const char* flag;     // command-line spelling or nullptr
  const char* section;  // config.json section under "cpp" / "common", or nullptr
  const char* key;
  Slot slot;
  bool set_on_cli = false;
};

std::vector<Row> setting_rows(Settings& s) {
  PipelineConfig& p = s.pipe;
  return {
      {"--config", nullptr, nullptr, {Slot::Path, &s.config_path}},
      {"--export-geometry", "outputs", "export_geometry", {Slot::GeometryName, &s.geometry}},
      {"--mesh-kf", "mesh_sparse", "kf", {Slot::Int, &s.mesh_kf}},
      {"--mesh-max-points", "mesh_sparse", "max_points", {Slot::Int, &s.mesh_max_points}},
      {"--mesh-grid-px", "mesh_sparse", "grid_px", {Slot::Int, &s.mesh_grid_px}},
      {"--mesh-max-edge-px", "mesh_sparse", "max_edge_px", {Slot::Real, &s.mesh_max_edge_px}},
      {nullptr, "system", "frames", {Slot::Int, &s.frames}},  // the command line gives it as the third positional
      {nullptr, "klt", "max_tracks", {Slot::Int, &p.klt.max_tracks}},
      {nullptr, "klt", "min_tracks", {Slot::Int, &p.klt.min_tracks}},
      {nullptr, "klt", "quality", {Slot::Real, &p.klt.quality}},
      {nullptr, "klt", "min_distance", {Slot::Int, &p.klt.min_distance}},
      {nullptr, "klt", "pyr_levels", {Slot::Int, &p.klt.pyr_levels}},
      {nullptr, "klt", "win_radius", {Slot::Int, &p.klt.win_radius}},
      {nullptr, "klt", "iters", {Slot::Int, &p.klt.iters}},
      {nullptr, "klt", "fb_thresh", {Slot::Real, &p.klt.fb_thresh}},
      {nullptr, "keyframe", "min_gap", {Slot::Int, &p.kf_min_gap}},
      {nullptr, "keyframe", "min_inliers", {Slot::Int, &p.kf_min_inliers}},
      {nullptr, "keyframe", "parallax_px", {Slot::Real, &p.kf_parallax_px}},
      {nullptr, "ba", "window", {Slot::Int, &p.ba.window}},
      {nullptr, "ba", "iters", {Slot::Int, &p.ba.iters}},
      {nullptr, "ba", "max_points", {Slot::Int, &p.ba.max_points}},
      {nullptr, "ba", "huber_delta", {Slot::Real, &p.ba.huber_delta}},
      {nullptr, "ba", "lambda", {Slot::Real, &p.ba.lambda}},
  };
}

void print_usage(const char* argv0) {  // text of T:1521-1534
  std::cerr << "Usage: " << argv0 << " <templering_root> <out_dir> [frames] [options]\n"
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
}

// Fills `s` from argv[3..] and the config file.  Returns false when --help ended the run (exit code 0).
bool gather_settings(int argc, char** argv, Settings& s) {
  std::vector<Row> rows = setting_rows(s);
  std::vector<std::string> args(argv + 3, argv + argc);
  size_t at = 0;
  bool frames_on_cli = false;
  if (!args.empty() && !args[0].empty() && args[0][0] != '-') {  // optional third positional: frames (T:1546-1553)
    s.frames = std::stoi(args[0]);
    frames_on_cli = true;
    at = 1;
  }
  while (at < args.size()) {
    const std::string& flag = args[at++];
    if (flag == "-h" || flag == "--help") {
      std::cerr << "Run without args to see usage.\n";
      return false;
    }
    Row* row = nullptr;
    for (Row& r : rows)
      if (r.flag && flag == r.flag) row = &r;
    if (!row) throw std::runtime_error("Unknown option: " + flag);
    if (at >= args.size()) throw std::runtime_error("Missing value for " + flag);
    row->slot.from_text(args[at++], flag);
    row->set_on_cli = true;
  }
  s.config_given = rows[0].set_on_cli;
  if (!s.config_given && fs::exists("config.json")) {  // auto-discovery in the working directory (T:1613-1619)
    s.config_path = "config.json";
    s.config_given = true;
  }
  if (!s.config_given) return true;
  Json doc;
  try {
    std::ifstream f(s.config_path);
    if (!f) throw std::runtime_error("Failed to open: " + s.config_path);
    std::ostringstream text;
    text << f.rdbuf();
    doc = Json::parse(text.str());
  } catch (const std::exception& e) {
    throw std::runtime_error("Failed to parse config.json: " + s.config_path + " | " + e.what());
  }
  for (const Row& r : rows) {
    if (!r.section || r.set_on_cli) continue;
    if (r.slot.at == &s.frames) {
      if (frames_on_cli) continue;
      if (const auto v = as_int(config_value(doc, r.section, r.key))) s.frames = std::max(1, *v);  // T:1633-1635
      continue;
    }
    r.slot.from_config(config_value(doc, r.section, r.key));
  }
  return true;
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
      print_usage(argv[0]);
      return 2;
    }
    const fs::path root = fs::path(argv[1]);
    const fs::path out = fs::path(argv[2]);
    Settings st;
    if (!gather_settings(argc, argv, st)) return 0;
    PipelineConfig& pc = st.pipe;
    const int frames = st.frames;
    pc.frames = frames;
    pc.export_pointcloud = st.geometry == Geometry::PointCloud || st.geometry == Geometry::Both;
    const bool want_mesh = st.geometry == Geometry::Mesh || st.geometry == Geometry::Both;

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

    // Multi-GPU mode (not part of the reference's command line, so it lives in the environment): SFMX_DIST_WORLD=N processes,
    // one per GPU, each started with its SFMX_DIST_RANK and a common SFMX_DIST_ID_FILE, run the SAME sequence and shard BA
    // points and RANSAC hypotheses.  Rank 0 writes the RCCL unique ids into the file, prints and writes outputs.
    const int dist_world = std::getenv("SFMX_DIST_WORLD") ? std::atoi(std::getenv("SFMX_DIST_WORLD")) : 1;
    const int dist_rank = std::getenv("SFMX_DIST_RANK") ? std::atoi(std::getenv("SFMX_DIST_RANK")) : 0;
    const char* dev_env = std::getenv("SFMX_DEVICE");
    const int device = dev_env ? std::atoi(dev_env) : (dist_world > 1 ? dist_rank : 0);
    sfmx_ctx* ctx = nullptr;
    const int rc = sfmx_ctx_create(device, &ctx);
    if (rc != SFMX_OK) throw std::runtime_error("no usable MI355X (gfx950) device: sfmx_ctx_create failed (there is no CPU fallback)");
    struct CtxGuard { sfmx_ctx* c; ~CtxGuard() { sfmx_ctx_destroy(c); } } guard{ctx};
    struct Comms {
      sfmx_comm* c[2] = {nullptr, nullptr};  // BA lane, RANSAC merges of the geometry thread (pipeline.hpp: PipelineConfig)
      ~Comms() { for (sfmx_comm* m : c) sfmx_comm_destroy(m); }
    } comms;
    if (dist_world > 1) {
      const char* idf = std::getenv("SFMX_DIST_ID_FILE");
      if (!idf || dist_rank < 0 || dist_rank >= dist_world) throw std::runtime_error("SFMX_DIST_WORLD needs SFMX_DIST_RANK in range and SFMX_DIST_ID_FILE");
      // File = 16-byte header {"SFMXID02", run id} + the unique ids.  The run id (FNV-1a of SFMX_DIST_RUN_ID, which the launcher
      // sets to something fresh per launch) keeps a rank from picking up the file a PREVIOUS launch left at the same path --
      // mismatched ids would block ncclCommInitRank forever.  Without a run id a file older than the waiting time is refused.
      std::uint64_t run_id = 0;
      if (const char* rid = std::getenv("SFMX_DIST_RUN_ID")) {
        run_id = 1469598103934665603ull;
        for (const char* q = rid; *q; ++q) run_id = (run_id ^ (unsigned char)*q) * 1099511628211ull;
        if (run_id == 0) run_id = 1;
      }
      constexpr int kComms = 2;
      constexpr size_t kHeader = 16;
      std::string blob(kHeader + (size_t)kComms * SFMX_COMM_ID_BYTES, '\0');
      std::memcpy(&blob[0], "SFMXID02", 8);
      std::memcpy(&blob[8], &run_id, 8);
      if (dist_rank == 0) {
        std::error_code ec;
        fs::remove(idf, ec);  // whatever an earlier launch left behind
        for (int k = 0; k < kComms; k++)
          if (sfmx_comm_get_unique_id(&blob[kHeader + (size_t)k * SFMX_COMM_ID_BYTES]) != SFMX_OK) throw std::runtime_error("RCCL is not available (sfmx_comm_get_unique_id)");
        const std::string tmp = std::string(idf) + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write(blob.data(), (std::streamsize)blob.size()); }
        fs::rename(tmp, idf);
      } else {
        const auto started = fs::file_time_type::clock::now();
        std::string got(blob.size(), '\0');
        for (int tries = 0;; ++tries) {  // wait for rank 0 (up to two minutes)
          std::ifstream f(idf, std::ios::binary);
          bool ok = f && f.read(&got[0], (std::streamsize)got.size()) && std::memcmp(got.data(), blob.data(), kHeader) == 0;
          if (ok && run_id == 0) {  // no run id to tell launches apart: a file older than the two minutes a rank waits is stale
            std::error_code ec;
            const auto mt = fs::last_write_time(idf, ec);
            ok = !ec && mt + std::chrono::seconds(120) >= started;
          }
          if (ok) break;
          if (tries > 1200) throw std::runtime_error(std::string("timed out waiting for ") + idf + " (missing, stale, or written for another SFMX_DIST_RUN_ID)");
          std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        blob = got;
      }
      for (int k = 0; k < kComms; k++)
        if (sfmx_comm_create(device, &blob[kHeader + (size_t)k * SFMX_COMM_ID_BYTES], dist_rank, dist_world, &comms.c[k]) != SFMX_OK)
          throw std::runtime_error("sfmx_comm_create failed (RCCL)");
      pc.comm_ba = comms.c[0];
      pc.comm_ransac = comms.c[1];
    }
    const bool speaker = dist_rank == 0;  // every rank computes the same result; one of them reports it

    PipelineResult res;
    run_pipeline(ctx, src, meta, K, pc, res, speaker ? echo_line : nullptr);
    if (!speaker) return 0;
    const size_t printed = res.log.size();
    write_outputs(out.string(), pc, meta, res);
    if (want_mesh) {  // T:1884-1906
      if (res.kfs.empty()) {
        std::cerr << "WARN: mesh export skipped (no keyframes).\n";
      } else {
        const int kidx = std::max(0, std::min(st.mesh_kf, (int)res.kfs.size() - 1));
        const Keyframe& mkf = res.kfs[(size_t)kidx];
        const Gray im = read_pgm((src.dir / fs::path(mkf.img_name).replace_extension(".pgm")).string());  // its size bounds the projection
        std::vector<V3> verts;
        std::vector<std::array<int, 3>> faces;
        build_sparse_mesh(K, mkf.pose, res.map, im.w, im.h, st.mesh_max_points, st.mesh_grid_px, st.mesh_max_edge_px, verts, faces);
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
