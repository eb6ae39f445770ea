// pipeline.hpp — C++20 host side of the SfM hot path: the reference's function seams
// (KLTTracker::step, shi_tomasi, find_E_ransac, bundle_adjust_window) re-hosted on
// the sfmx C ABI, plus the per-frame loop of main() (T:1686-1911) and its CSV/PLY writers.
// Nothing here falls back to CPU arithmetic for the hot kernels: a failing C-ABI call is an error.
#pragma once
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <optional>
#include <thread>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/sfmx.h"
#include "arena.hpp"
#include "host_math.hpp"
#include "introsort_replay.hpp"

namespace sfmx_host {

struct SfmxFailure : std::runtime_error {
  int status;
  SfmxFailure(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// T:307-316
struct LKConfig {
  int max_tracks = 2200, min_tracks = 900;
  double quality = 0.01;
  int min_distance = 8, pyr_levels = 3, win_radius = 5, iters = 10;
  double fb_thresh = 1.0;
};
// T:811-817
struct BAConfig {
  int window = 6, iters = 5, max_points = 600;
  double huber_delta = 3.0, lambda = 1e-3;
};
struct Track { int id; V2 p; };
struct StepOut { std::vector<V2> prev_pts, cur_pts; std::vector<int> ids; };

// where frames come from: host pixels, or pixels already resident in HBM
struct FrameSource {
  virtual ~FrameSource() = default;
  virtual int count() const = 0;
  virtual int width() const = 0;
  virtual int height() const = 0;
  // loads frame fi into pyr (level 0 + downsampled levels); throws std::runtime_error like read_pgm
  virtual void load(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr) = 0;
  // the same ahead of time on the context's second stream (sfmx_pyramid_set_device_async), if the source can: false = not
  // started, the caller loads the frame when it needs it.  fetch_level >= 0: that level's pixels also travel to the host.
  virtual bool load_async(sfmx_ctx*, int, sfmx_pyramid*, int /*fetch_level*/) { return false; }
};
struct MemoryFrames : FrameSource {
  const std::uint8_t* host = nullptr;  // [n][h][w]
  const std::uint8_t* dev = nullptr;   // same layout in HBM (preferred when non-null)
  int n = 0, w = 0, h = 0;
  int count() const override { return n; }
  int width() const override { return w; }
  int height() const override { return h; }
  void load(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr) override;
  bool load_async(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr, int fetch_level) override;
};

struct StageClock {
  double klt = 0, shi = 0, ransac = 0, ba = 0, upload = 0, host = 0, total = 0, shi_gpu = 0, shi_replay = 0, desc = 0, bookkeeping = 0;
  double r_pre = 0, r_gpu = 0, r_verify = 0, r_decomp = 0, tri_iter = 0, tri_solve = 0, tri_insert = 0;
  double klt_kernel_us = 0, ransac_kernel_us = 0, ba_kernel_us = 0, shi_kernel_us = 0;
  std::uint64_t lk_steps = 0, tracks_in = 0, ransac_calls = 0, ransac_points = 0, ba_calls = 0, ba_iters = 0, klt_calls = 0;
  std::uint64_t ransac_cert_misses = 0;  // a certified inlier count that did not hold (parity fallback taken)
  std::uint64_t ransac_verified = 0, shi_fallbacks = 0, shi_calls = 0, shi_memo_hits = 0, shi_prefetched = 0;
  double shi_wait = 0, setup = 0;
  double pf_busy = 0, pf_gpu = 0, pf_replay = 0, lane_a_busy = 0, lane_b_busy = 0, lane_c_busy = 0, lane_e_busy = 0, join_wait = 0, ba_gather = 0;
  double m_step = 0, m_ransac = 0, m_kf = 0;  // wall time of tracker.step (tracker lane) / frame->frame RANSAC / keyframe block
  double feed_wait = 0;                        // geometry lane waiting for the tracker lane
  // per-kernel GPU time / launches over every context of the run (sfmx_kernel_profile; only with timing enabled)
  static constexpr int kKernels = 16;
  double kernel_us[kKernels] = {};
  std::uint64_t kernel_calls[kKernels] = {};
  void grab_profile(sfmx_ctx* ctx);            // adds (and clears) the context's profile
  void add(const StageClock& o);               // field-wise sum (lane clocks are folded into the run's clock)
};

// shi_tomasi (T:237-302) on one context: device score + certain-outcome fixpoint (sfmx_shi_tomasi_candidates_pruned),
// then the reference's sort + greedy min-distance pick on the few surviving candidates (tie order via the
// introsort replay when needed).  One detector per context / per thread.
// Helper contexts (corner prefetch workers, lanes B/C) are recycled across pipeline runs of one process: creating and
// destroying a context -- streams, hipMalloc/hipHostMalloc'ed slabs that grow on first use, captured graphs -- costs
// several milliseconds per run, more than a tenth of a 47-frame pass.  Entries are leaked at process exit on purpose
// (no HIP calls from static destructors); sfmx_host_release_contexts() frees them explicitly.
struct PooledCtx {
  sfmx_ctx* ctx = nullptr;
  int device = 0, priority = 0;
  int role = 0;                 // which lane it serves: the same lane gets the same context (and hardware queue) every run
  sfmx_pyramid* pyr = nullptr;  // one scratch pyramid that lives with the context
  int pw = 0, ph = 0, pl = 0;
  sfmx_pyramid* pyramid(int w, int h, int levels);  // (re)created when the geometry changes
  std::vector<sfmx_pyramid*> ring;  // the tracker lane's pyramids, same life cycle
  int rw = 0, rh = 0, rl = 0;
  const std::vector<sfmx_pyramid*>& pyramid_ring(int w, int h, int levels, int count);
  sfmx_ba_problem* ba = nullptr;  // lane B's BA problem object (grow-only device buffers), same life cycle
  void free_pyramids();
};
class ContextPool {
 public:
  static ContextPool& instance();
  enum Role { PREFETCH = 1, TRACKER = 2, LANE_B = 3, LANE_C = 4, LANE_A = 5, LANE_E = 6, LANE_A2 = 7 };
  PooledCtx* acquire(int device, int priority, int role);
  void release(PooledCtx* pc);  // synchronises the context; the caller's threads must have stopped using it
  void clear();

 private:
  std::mutex mu_;
  std::vector<PooledCtx*> free_;
};

// A detection whose tie order still has to be resolved on the host (no device access needed any more): the survivors
// of the device fixpoint plus a private copy of the sort keys of ALL candidates.
struct CornerTies {
  int w = 0, h = 0, min_dist = 0, max_corners = 0, n_total = 0;
  std::vector<std::uint32_t> xy;       // survivors: x | y<<16 (| certain-accept flag in bit 31)
  std::vector<double> s;               // survivors: score
  std::vector<std::int32_t> full;      // survivors: index in the full candidate list
  std::vector<SortKey> keys;           // all candidates, row-major (the reference's push order, T:282-284)
};

class CornerDetector {
 public:
  CornerDetector(sfmx_ctx* ctx, StageClock* clk) : ctx_(ctx), clk_(clk) {}
  std::vector<V2> detect(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist);
  // Two-phase form used by the prefetcher.  detect_device: device score + fixpoint and the tie-free host walk;
  // returns true when `out` is final, false when `ties` was filled and resolve_ties() has to finish the job.
  bool detect_device(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist, std::vector<V2>& out, CornerTies& ties);
  // pure host code (any thread, no context): introsort replay + greedy pick.  false = replay declined (the caller
  // falls back to detect(), i.e. the reference's own sort call on the full list).
  static bool resolve_ties(CornerTies& ties, std::vector<V2>& out, StageClock* clk);

 private:
  std::vector<V2> detect_full_sort(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist);
  sfmx_ctx* ctx_;
  StageClock* clk_;
  std::vector<std::uint32_t> cand_xy_;
  std::vector<double> cand_s_;
  std::vector<std::int32_t> cand_full_;
  CornerTies ties_;  // scratch of the synchronous detect()
};

// Corner detection of frame f+1 depends only on image f+1, so it is computed ahead of time by a worker thread
// with its OWN sfmx context (own HIP stream and buffers) while the main thread tracks / scores / adjusts frame f.
// The result is the uncapped accepted-corner sequence; GpuTracker::shi_tomasi takes prefixes of it.
class CornerPrefetcher {
 public:
  CornerPrefetcher(int device, FrameSource& src, double quality, int min_dist, int workers, bool timing);
  ~CornerPrefetcher();
  CornerPrefetcher(const CornerPrefetcher&) = delete;
  CornerPrefetcher& operator=(const CornerPrefetcher&) = delete;
  void request(int frame);                          // non-blocking; ignored if already requested
  bool take(int frame, std::vector<V2>& corners);   // waits for a requested frame; false if never requested / failed
  bool take_if_done(int frame, std::vector<V2>& corners);  // the same without waiting: false if not (yet) there
  double quality() const { return quality_; }
  int min_dist() const { return min_dist_; }
  void discard_older_than(int frame);               // drop finished results of frames < frame that nobody took
  bool matches(double quality, int min_dist) const { return quality == quality_ && min_dist == min_dist_; }
  double kernel_us();                               // accumulated device time of the workers' score kernels
  void grab_profile(StageClock& clk);               // per-kernel profiles of the workers' contexts (call after the run)
  void busy(double& total, double& gpu, double& replay);  // seconds over all workers (call after the run)
  std::uint64_t replays();                          // tie-order replays over all workers (call after the run)

 private:
  // one worker = a device thread (own context + detector) and a resolver thread (host-only tie resolution of the
  // previous image while the device thread is already on the next one); frames come from a shared queue
  struct Worker {
    PooledCtx* pc = nullptr;
    sfmx_ctx* ctx = nullptr;      // = pc->ctx
    sfmx_pyramid* pyr = nullptr;  // = pc's scratch pyramid
    std::unique_ptr<CornerDetector> det;
    StageClock clock, clock_resolver;
    double busy = 0, busy_resolver = 0;
    std::thread th, th_resolver;
    // tie hand-over between the two threads (all guarded by the prefetcher's mutex)
    CornerTies ties[2];
    std::vector<CornerTies*> free_ties;
    std::deque<std::pair<int, CornerTies*>> ties_queue;
    std::condition_variable cv_ties;
  };
  void run(Worker& w);
  void run_resolver(Worker& w);
  void publish(int frame, std::vector<V2>&& seq, bool failed);
  void shutdown();
  struct Slot { bool done = false, failed = false; std::vector<V2> corners; };
  FrameSource& src_;
  double quality_;
  int min_dist_;
  std::vector<std::unique_ptr<Worker>> workers_;
  std::mutex mu_;
  std::condition_variable cv_req_, cv_done_;
  std::deque<int> queue_;
  std::unordered_map<int, Slot> slots_;
  bool stop_ = false;
};

// The accepted-corner sequence of one image: the greedy pick accepts candidates in a fixed order and max_corners only
// truncates it, so any request with the same quality / min_dist and max_corners <= cap (or any, if the pick ran to
// exhaustion) is a prefix of it.
struct CornerMemo {
  double quality;
  int min_dist;
  int cap;
  bool exhausted;
  std::vector<V2> corners;
  bool serves(int max_corners, double q, int md) const { return max_corners >= 1 && quality == q && min_dist == md && (exhausted || max_corners <= cap); }
  std::vector<V2> prefix(int max_corners) const {
    return std::vector<V2>(corners.begin(), corners.begin() + (long)std::min((size_t)std::max(max_corners, 0), corners.size()));
  }
};

// KLTTracker (T:323-466) on the GPU
class GpuTracker {
 public:
  // ring >= 2 pyramids: the k-th processed frame lives in slot k % ring.  before_load(fi), if set, is called before
  // frame fi overwrites the slot of frame fi - ring (the tracker lane waits there until that frame is released).
  // borrowed: pyramids owned by somebody else (a pooled context); otherwise `ring` pyramids are created and owned here
  GpuTracker(sfmx_ctx* ctx, LKConfig cfg, int w, int h, int extra_levels, StageClock* clk, int ring = 2,
             std::function<void(int)> before_load = nullptr, const std::vector<sfmx_pyramid*>* borrowed = nullptr);
  // Preloading: while the KLT launch of frame fi runs, the pyramid of frame fi + 1 is built on the context's second stream
  // (and the pixels of `fetch_level`, the descriptor's source, copied to the host).  may_load(f) must say -- without
  // blocking -- whether the ring slot of frame f is free already.
  void enable_preload(std::function<bool(int)> may_load, int fetch_level) { may_load_ = std::move(may_load); fetch_level_ = fetch_level; }
  ~GpuTracker();
  GpuTracker(const GpuTracker&) = delete;
  GpuTracker& operator=(const GpuTracker&) = delete;
  StepOut step(FrameSource& src, int fi);
  const std::vector<Track>& tracks() const { return tracks_; }
  sfmx_pyramid* current() const { return ring_[(size_t)slot_]; }  // pyramid of the most recent frame
  const LKConfig& cfg() const { return cfg_; }
  // shi_tomasi (T:237-302): device score + certain-outcome fixpoint, host sort + greedy pick on the survivors.
  // frame_key >= 0 memoises the accepted-corner sequence of that frame (CornerMemo).
  std::vector<V2> shi_tomasi(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist, int frame_key = -1);
  // hands the memo of a frame over (the loop-closure verification re-detects corners on old keyframe images, T:1841)
  std::shared_ptr<const CornerMemo> take_memo(int frame_key);
  void set_prefetcher(CornerPrefetcher* p) { prefetch_ = p; }
  int levels_total() const { return levels_total_; }

 private:
  void reset(FrameSource& src, int fi);
  sfmx_ctx* ctx_;
  LKConfig cfg_;
  int w_, h_, levels_total_;
  std::vector<sfmx_pyramid*> ring_;
  bool owns_ring_ = true;
  int slot_ = 0;
  std::function<void(int)> before_load_;
  std::function<bool(int)> may_load_;
  int fetch_level_ = -1, preloaded_frame_ = -1, preloaded_slot_ = -1;
  bool have_prev_ = false;
  std::vector<Track> tracks_;
  std::vector<int> grid_head_, grid_next_;  // scratch of the replenish distance filter
  int next_id_ = 0;
  StageClock* clk_;
  CornerDetector det_;
  CornerPrefetcher* prefetch_ = nullptr;
  std::unordered_map<int, CornerMemo> corner_cache_;
};

struct RelPose {
  Mat3 R_ji;
  V3 t_ji;
  std::vector<int> inliers;
  int best_iter = -1;
};

// One rank's half of a find_E_ransac call (T:646-761): the exact winner of ITS contiguous range of the common iteration
// stream (ransac_local).  ransac_merge turns the ranks' local winners into the call's result; with one rank it only
// adds the decomposition (T:680-760).
struct RansacLocal {
  bool none = false;               // fewer than 8 correspondences (T:648) or no iterations: the call returns nullopt
  int n = 0, min_inliers = 0;
  double thr = 0.0;
  std::vector<double> xi, xj;      // K^-1-normalised correspondences [n][2]
  int win_iter = -1, win_count = -1;
  Mat3 winE;
  std::vector<std::uint8_t> win_mask;
};
// A RANSAC call that ran ahead of the geometry thread: finished (rel) or, in a multi-GPU run, waiting for its merge
struct RansacAhead {
  bool merged = true;
  std::optional<RelPose> rel;
  RansacLocal local;
};

// What the geometry lane needs of one frame.  pyr stays valid until FrameFeeder::release_upto(fi).
struct FramePacket {
  int fi = -1;
  StepOut step;                                  // empty prev_pts: the tracker (re)started on this frame (T:341-344)
  std::vector<Track> tracks;                     // live tracks after the step, replenished ones included
  const sfmx_pyramid* pyr = nullptr;             // this frame's pyramid (tracker context, same device)
  std::vector<float> desc;                       // global_desc_32 of the frame (T:1100-1122)
  std::shared_ptr<const CornerMemo> corners;     // accepted-corner sequence, if this frame's image was detected
  // frame->frame find_E_ransac of this step (T:1739), when it was started ahead of the geometry lane (lane A): a pure
  // function of step.prev_pts / cur_pts (RNG seeded inside, T:657).  get() rethrows what it threw ("Singular K").
  std::shared_future<std::shared_ptr<RansacAhead>> rel;
};

// Tracker lane.  KLTTracker::step depends on the images and on its own previous state only -- never on poses, keyframe
// decisions or the map -- so frame f+1.. are tracked on a context of their own while the geometry lane (RANSAC,
// keyframes, triangulation, BA hand-over) is still busy with frame f.  The ring of pyramids bounds how far it runs
// ahead; inline mode (threaded = false) produces each packet inside next() on the caller's context.
class FrameFeeder {
 public:
  // on_packet (optional) runs on the producing thread right after a packet is complete, before it is queued
  FrameFeeder(sfmx_ctx* caller_ctx, FrameSource& src, const LKConfig& cfg, int extra_levels, int desc_level, int n_frames, bool threaded,
              CornerPrefetcher* prefetch, int prefetch_depth, StageClock* clk, std::function<void(FramePacket&)> on_packet = nullptr);
  ~FrameFeeder();
  FrameFeeder(const FrameFeeder&) = delete;
  FrameFeeder& operator=(const FrameFeeder&) = delete;
  FramePacket next();             // packets arrive in frame order; rethrows what the tracker lane threw
  void release_upto(int frame);   // the pyramids of frames <= frame may be overwritten
  void finish();                  // waits for the lane to end (all packets produced); its clock may be read afterwards
  int levels_total() const { return tracker_->levels_total(); }
  bool threaded() const { return pc_ != nullptr; }
  sfmx_ctx* ctx() const { return ctx_; }
  StageClock& lane_clock() { return lane_clk_; }  // the tracker lane's own counters (threaded mode; read after the run)

 private:
  FramePacket produce(int fi);
  void run();
  FrameSource& src_;
  int desc_level_, n_frames_, next_frame_ = 0, ring_ = 2, prefetch_depth_;
  CornerPrefetcher* prefetch_;
  std::function<void(FramePacket&)> on_packet_;
  PooledCtx* pc_ = nullptr;
  sfmx_ctx* ctx_;
  StageClock lane_clk_;
  StageClock* clk_;
  std::unique_ptr<GpuTracker> tracker_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_pkt_, cv_rel_;
  std::deque<FramePacket> queue_;
  int released_ = -1;
  std::atomic<int> released_a_{-1};           // copies for the short polls that precede the sleeps on the condition variables
  std::atomic<std::uint64_t> produced_{0};
  std::uint64_t consumed_ = 0;
  bool stop_ = false, done_ = false;
  std::exception_ptr error_;
};

// fwd/bwd track of arbitrary points between two pyramids on any context of the same device (loop-closure verification)
void klt_pairs(sfmx_ctx* ctx, const LKConfig& cfg, const sfmx_pyramid* a, const sfmx_pyramid* b, const std::vector<V2>& p0, std::vector<V2>& fwd,
               std::vector<std::uint8_t>& keep, StageClock* clk);

// find_E_ransac (T:646-761); throws std::runtime_error("Singular K") like the reference
// comm (optional): the iterations are sharded over its ranks; counts are exact per iteration, so every rank returns the
// single-GPU result bit for bit (all-reduce(max) of the packed (count, iteration) key, winner's E carried as raw bits)
RansacLocal ransac_local(sfmx_ctx* ctx, const Mat3& K, const std::vector<V2>& pi, const std::vector<V2>& pj, int iters, double thr,
                         int min_inliers, StageClock* clk, int rank, int world);
std::optional<RelPose> ransac_merge(sfmx_ctx* ctx, sfmx_comm* comm, RansacLocal&& local, StageClock* clk);
std::optional<RelPose> find_E_ransac_gpu(sfmx_ctx* ctx, const Mat3& K, const std::vector<V2>& pi, const std::vector<V2>& pj, int iters,
                                         double thr, int min_inliers, StageClock* clk, sfmx_comm* comm = nullptr);

// T:766-798
struct Keyframe {
  int kf_id = 0, frame_idx = 0;
  std::string img_name;
  Pose pose;
  ArenaMap<int, V2> obs;
  Keyframe() = default;
  explicit Keyframe(Arena* a) : obs(0, std::hash<int>(), std::equal_to<int>(), ArenaAlloc<std::pair<const int, V2>>(a)) {}
};
struct MapPoint {
  int pid = 0, tid = 0;
  V3 Xw;
  std::vector<std::pair<int, V2>, ArenaAlloc<std::pair<int, V2>>> obs;  // in the map's arena (null arena: plain new/delete)
  explicit MapPoint(Arena* a = nullptr) : obs(ArenaAlloc<std::pair<int, V2>>(a)) {}
};
struct MapState {
  int next_pid = 0;
  ArenaMap<int, int> tid2pid;
  ArenaMap<int, MapPoint> pts;
  MapState() = default;
  explicit MapState(Arena* a)
      : tid2pid(0, std::hash<int>(), std::equal_to<int>(), ArenaAlloc<std::pair<const int, int>>(a)),
        pts(0, std::hash<int>(), std::equal_to<int>(), ArenaAlloc<std::pair<const int, MapPoint>>(a)) {}
  bool has(int tid) const { return tid2pid.find(tid) != tid2pid.end(); }
  int add(int tid, V3 Xw);
  void add_obs(int tid, int kf_id, V2 uv);
};
struct PGEdge {
  int i = -1, j = -1;
  Mat3 R_ji;
  V3 t_ji;
  int inliers = 0;
  bool is_loop = false;
};

// bundle_adjust_window (T:848-1097) in three steps so that the GPU part can run on another lane:
//   gather (host, touches kfs/map): window selection + the reference's point collection order (T:853-883);
//   solve  (device + SO(3) updates on private pose copies): the iteration loop (T:893-1096);
//   apply  (host): write the refined poses back into kfs[w0+1..] (T:1093-1094).
struct BaJob {
  bool valid = false;
  int w0 = 0, W = 0, P = 0;
  Mat3 K;
  BAConfig cfg;
  std::vector<double> X, uv;
  std::vector<std::int32_t> optr, oli;
  std::vector<Pose> win;  // the window's camera->world poses; refined in place by solve()
};
class GpuBundleAdjuster {
 public:
  // comm (optional): the window's points are sharded over its ranks, S | b all-reduced per iteration (sfmx_ba_step_sharded)
  // keep (optional): where the problem object lives between runs (a pooled context); otherwise it is destroyed with this
  GpuBundleAdjuster(sfmx_ctx* ctx, StageClock* clk, sfmx_comm* comm = nullptr, sfmx_ba_problem** keep = nullptr)
      : ctx_(ctx), clk_(clk), comm_(comm), prob_(keep ? *keep : nullptr), keep_(keep) {}
  ~GpuBundleAdjuster();
  static BaJob gather(const Mat3& K, const std::vector<Keyframe>& kfs, const MapState& map, const BAConfig& cfg);
  void solve(BaJob& job);
  static void apply(const BaJob& job, std::vector<Keyframe>& kfs);
  void run(const Mat3& K, std::vector<Keyframe>& kfs, MapState& map, const BAConfig& cfg) {
    BaJob j = gather(K, kfs, map, cfg);
    solve(j);
    apply(j, kfs);
  }

 private:
  sfmx_ctx* ctx_;
  StageClock* clk_;
  sfmx_comm* comm_ = nullptr;
  sfmx_ba_problem* prob_ = nullptr;
  sfmx_ba_problem** keep_ = nullptr;
};

// A second execution lane: one worker thread with its OWN sfmx context (own HIP stream and device buffers).
// Tasks run in submission order; wait() blocks until the lane is idle and rethrows the first task exception.
class AsyncLane {
 public:
  AsyncLane(int device, int priority, int role);  // priority: sfmx_ctx_create_prio; role: ContextPool::Role
  ~AsyncLane();
  AsyncLane(const AsyncLane&) = delete;
  AsyncLane& operator=(const AsyncLane&) = delete;
  sfmx_ctx* ctx() const { return ctx_; }
  PooledCtx* pooled() const { return pc_; }
  void submit(std::function<void()> task);
  // the same, returning the task's ticket: wait_ticket(t) returns once that task (and every task before it) has finished --
  // tasks submitted after it may still be running
  std::uint64_t submit_ticket(std::function<void()> task);
  void wait_ticket(std::uint64_t ticket);
  void wait();
  double busy_seconds() const { return busy_seconds_; }  // time spent inside tasks (read while idle)
  // this lane's own polling window before it (and whoever waits for it) sleeps; -1 = the process-wide SFMX_SPIN_US
  void set_spin_us(int us) { spin_us_.store(us, std::memory_order_relaxed); }

 private:
  void run();
  std::atomic<int> spin_us_{-1};
  double busy_seconds_ = 0;
  PooledCtx* pc_ = nullptr;
  sfmx_ctx* ctx_ = nullptr;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_task_, cv_idle_;
  std::deque<std::function<void()>> queue_;
  bool busy_ = false, stop_ = false;
  // hand-offs of a few hundred microseconds apart: both sides poll these for a short while before they sleep on the condition
  // variables (a futex wake-up costs 30-60 us per direction, twice per BA job and RANSAC call)
  std::atomic<std::uint64_t> submitted_{0}, completed_{0};
  std::exception_ptr error_;
};

bool posegraph_optimize_centers(sfmx_ctx* ctx, std::vector<Keyframe>& kfs, const std::vector<PGEdge>& edges);

// optional sparse-mesh export of the CLI (mesh.cpp; T:1226-1461): empty outputs = skipped
void build_sparse_mesh(const Mat3& K, const Pose& kf_pose, const MapState& map, int img_w, int img_h, int max_points, int grid_px,
                       double max_edge_px, std::vector<V3>& vertices, std::vector<std::array<int, 3>>& faces);
void write_mesh_ply(const std::string& path, const std::vector<V3>& vertices, const std::vector<std::array<int, 3>>& faces);

struct PipelineConfig {
  int frames = 12;
  bool export_pointcloud = true;
  LKConfig klt;
  BAConfig ba;
  int kf_min_gap = 1, kf_min_inliers = 200;
  double kf_parallax_px = 18.0;
  // Multi-GPU mode (one process per GPU, every rank runs the same sequence): BA points and RANSAC hypotheses are
  // sharded over the ranks of these communicators.  TWO communicators, each used by exactly one thread at a
  // time and in an order that is the same on every rank:
  //   comm_ba      the S | b all-reduce of every BA iteration; issued by lane B, job after job in keyframe order (and by the
  //                geometry thread for the second BA of an accepted loop closure, while lane B is idle);
  //   comm_ransac  the winner merge of every find_E_ransac call; issued by the GEOMETRY thread where it consumes the call
  //                (program order), whichever lane scored the rank's share of the hypotheses ahead of time.
  // null = unsharded.
  sfmx_comm* comm_ba = nullptr;
  sfmx_comm* comm_ransac = nullptr;
};
struct FrameMeta {
  std::string name;
  double lat = 0, lon = 0;
  bool has_ang = false;
};
struct PipelineResult {
  std::unique_ptr<Arena> arena = std::make_unique<Arena>();  // declared first: outlives the maps below
  std::vector<Keyframe> kfs;
  std::vector<PGEdge> edges;
  MapState map{arena.get()};
  std::string log;  // exactly what the reference prints to stdout
  StageClock clock;
};
// The per-frame loop of main() (T:1708-1871).  `echo` (optional) receives each stdout line as it is produced.
void run_pipeline(sfmx_ctx* ctx, FrameSource& src, const std::vector<FrameMeta>& meta, const Mat3& K, const PipelineConfig& cfg,
                  PipelineResult& out, void (*echo)(const std::string&) = nullptr);
// writers (T:1199-1224, 1463-1475) + the summary block (T:1908-1911); appends the summary to out.log
void write_outputs(const std::string& out_dir, const PipelineConfig& cfg, const std::vector<FrameMeta>& meta, PipelineResult& out);

}  // namespace sfmx_host
