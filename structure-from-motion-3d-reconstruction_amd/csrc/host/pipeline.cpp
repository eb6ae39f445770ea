// pipeline.cpp — see pipeline.hpp.  Reference citations: T: = cpp/src/templering_sfm.cpp.
#include "pipeline.hpp"

#include "../hip/sfmx_math.h"
#include "cli_io.hpp"
#include "introsort_replay.hpp"
#include "thread_pool.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>

namespace sfmx_host {

namespace {
using Clock = std::chrono::steady_clock;
inline double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

void check(sfmx_ctx* ctx, int rc, const char* where) {
  if (rc != SFMX_OK) throw SfmxFailure(rc, std::string("sfmx: ") + where + ": " + (ctx ? sfmx_last_error(ctx) : "no context"));
}
}  // namespace

void MemoryFrames::load(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr) {
  if (fi < 0 || fi >= n) throw std::runtime_error("frame index out of range");
  const size_t off = (size_t)fi * (size_t)w * (size_t)h;
  if (dev) check(ctx, sfmx_pyramid_set_device(ctx, pyr, dev + off), "pyramid_set_device");
  else check(ctx, sfmx_pyramid_upload(ctx, pyr, host + off), "pyramid_upload");
}

bool MemoryFrames::load_async(sfmx_ctx* ctx, int fi, sfmx_pyramid* pyr, int fetch_level) {
  if (!dev || fi < 0 || fi >= n) return false;
  check(ctx, sfmx_pyramid_set_device_async(ctx, pyr, dev + (size_t)fi * (size_t)w * (size_t)h, fetch_level), "pyramid_set_device_async");
  return true;
}

// ------------------------------------------------------------------------------------------ tracker
GpuTracker::GpuTracker(sfmx_ctx* ctx, LKConfig cfg, int w, int h, int extra_levels, StageClock* clk, int ring, std::function<void(int)> before_load,
                       const std::vector<sfmx_pyramid*>* borrowed)
    : ctx_(ctx), cfg_(cfg), w_(w), h_(h), before_load_(std::move(before_load)), clk_(clk), det_(ctx, clk) {
  levels_total_ = std::max(cfg.pyr_levels, extra_levels);
  if (borrowed) {
    ring_ = *borrowed;
    owns_ring_ = false;
    return;
  }
  try {
    for (int i = 0; i < std::max(2, ring); i++) {
      sfmx_pyramid* p = nullptr;
      check(ctx_, sfmx_pyramid_create(ctx_, w, h, levels_total_, &p), "pyramid_create");
      ring_.push_back(p);
    }
  } catch (...) {
    for (sfmx_pyramid* p : ring_) sfmx_pyramid_destroy(ctx_, p);
    throw;
  }
}
GpuTracker::~GpuTracker() {
  if (owns_ring_)
    for (sfmx_pyramid* p : ring_) sfmx_pyramid_destroy(ctx_, p);
}

std::shared_ptr<const CornerMemo> GpuTracker::take_memo(int frame_key) {
  auto it = corner_cache_.find(frame_key);
  if (it == corner_cache_.end()) return nullptr;
  auto m = std::make_shared<const CornerMemo>(std::move(it->second));
  corner_cache_.erase(it);
  return m;
}

std::vector<V2> GpuTracker::shi_tomasi(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist, int frame_key) {
  const auto t0 = Clock::now();
  if (clk_) clk_->shi_calls++;
  if (frame_key >= 0 && max_corners >= 1) {
    auto it = corner_cache_.find(frame_key);
    if (it != corner_cache_.end() && it->second.serves(max_corners, quality, min_dist)) {
      if (clk_) clk_->shi_memo_hits++;
      return it->second.prefix(max_corners);
    }
    if (prefetch_ && prefetch_->matches(quality, min_dist)) {  // computed ahead of time on the worker's context?
      std::vector<V2> seq;
      const auto tw = Clock::now();
      const bool ok = prefetch_->take(frame_key, seq);
      if (clk_) clk_->shi_wait += since(tw);
      if (ok) {
        if (clk_) clk_->shi_prefetched++;
        if (corner_cache_.size() >= 4096) corner_cache_.erase(corner_cache_.begin());
        auto& m = corner_cache_[frame_key];
        m = CornerMemo{quality, min_dist, 0x7fffffff, true, std::move(seq)};
        if (clk_) clk_->shi += since(t0);
        return m.prefix(max_corners);
      }
    }
  }
  std::vector<V2> out = det_.detect(pyr, max_corners, quality, min_dist);
  if (frame_key >= 0 && max_corners >= 1) {
    if (corner_cache_.size() >= 4096) corner_cache_.erase(corner_cache_.begin());  // bound the memo on very long runs
    corner_cache_[frame_key] = CornerMemo{quality, min_dist, max_corners, (int)out.size() < max_corners, out};
  }
  if (clk_) clk_->shi += since(t0);
  return out;
}

namespace {
struct Cand { int x, y; double s; int idx; };

// The accepted corners plus a uniform grid over them.  "Is c closer than min_dist to an accepted corner?" (T:292-296)
// is an existence test, so the grid answers it exactly like the reference's linear scan.
struct GreedyGrid {
  int cell, gw, gh;
  double md2;
  std::vector<int> head, next;
  std::vector<V2>& out;
  GreedyGrid(int w, int h, int min_dist, std::vector<V2>& o)
      : cell(std::max(1, min_dist)), gw(w / cell + 1), gh(h / cell + 1), md2((double)min_dist * min_dist), head((size_t)gw * gh, -1), out(o) {}
  bool blocked(const Cand& c) const {
    const int cx = c.x / cell, cy = c.y / cell;
    for (int gy = std::max(0, cy - 1); gy <= std::min(gh - 1, cy + 1); gy++)
      for (int gx = std::max(0, cx - 1); gx <= std::min(gw - 1, cx + 1); gx++)
        for (int idx = head[(size_t)gy * gw + gx]; idx >= 0; idx = next[(size_t)idx]) {
          const double dx = out[(size_t)idx].x - c.x, dy = out[(size_t)idx].y - c.y;
          if (dx * dx + dy * dy < md2) return true;
        }
    return false;
  }
  void accept(const Cand& c) {
    const int cx = c.x / cell, cy = c.y / cell;
    next.push_back(head[(size_t)cy * gw + cx]);
    head[(size_t)cy * gw + cx] = (int)out.size();
    out.push_back(V2{double(c.x), double(c.y)});
  }
  void greedy(const std::vector<Cand>& cands, int max_corners) {  // T:290-300 on an already ordered list
    for (const Cand& c : cands) {
      if (blocked(c)) continue;
      accept(c);
      if ((int)out.size() >= max_corners) break;
    }
  }
};

void unpack_survivors(const std::uint32_t* xy, const double* s, int n, std::vector<Cand>& cands) {
  cands.resize((size_t)n);
  for (int i = 0; i < n; i++) cands[(size_t)i] = {(int)(xy[i] & 0x7fffu), (int)((xy[i] >> 16) & 0x7fffu), s[i], i};
}

// Walk score groups of the survivors (sorted by descending score).  A group with a single member behaves as in the
// reference.  For a group of equal scores the reference's order is whatever its std::sort produced; the outcome is
// independent of that order iff at most ONE member is still eligible (not blocked by corners of strictly higher
// score) -- then exactly that member is accepted.  Two or more eligible members: ambiguous => false.
bool walk_tie_free(const std::vector<Cand>& cands, GreedyGrid& g, int max_corners) {
  size_t i = 0;
  while (i < cands.size() && (int)g.out.size() < std::max(1, max_corners)) {
    size_t j = i + 1;
    while (j < cands.size() && cands[j].s == cands[i].s) j++;
    if (j - i == 1) {
      if (!g.blocked(cands[i])) g.accept(cands[i]);
    } else {
      int eligible = -1, count = 0;
      for (size_t k = i; k < j; k++)
        if (!g.blocked(cands[k])) { count++; eligible = (int)k; }
      if (count >= 2) return false;
      if (count == 1) g.accept(cands[(size_t)eligible]);
    }
    i = j;
    if ((int)g.out.size() >= max_corners) break;
  }
  return true;
}
}  // namespace

bool CornerDetector::detect_device(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist, std::vector<V2>& out, CornerTies& ties) {
  const auto t0 = Clock::now();
  int w = 0, h = 0;
  sfmx_pyramid_level_size(pyr, 0, &w, &h);
  const int cap = w * h;
  static const bool force_full = std::getenv("SFMX_SHI_FULL_SORT") != nullptr;  // test hook: always take the exact slow path
  out.clear();
  if (force_full || min_dist < 1 || min_dist > 16) {
    out = detect_full_sort(pyr, max_corners, quality, min_dist);
    return true;
  }
  if ((int)cand_xy_.size() < cap) { cand_xy_.resize((size_t)cap); cand_s_.resize((size_t)cap); cand_full_.resize((size_t)cap); }
  // the device has already removed every candidate that is certainly rejected
  int n = 0, n_total = 0;
  double maxv = 0;
  check(ctx_, sfmx_shi_tomasi_candidates_pruned(ctx_, pyr, quality, min_dist, cap, cand_xy_.data(), cand_s_.data(), cand_full_.data(), &n,
                                                &n_total, &maxv),
        "shi_tomasi_candidates_pruned");
  if (clk_) { clk_->shi_kernel_us += sfmx_last_kernel_us(ctx_); clk_->shi_gpu += since(t0); }
  std::vector<Cand> cands;
  unpack_survivors(cand_xy_.data(), cand_s_.data(), n, cands);
  std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.s > b.s || (a.s == b.s && a.idx < b.idx); });
  out.reserve((size_t)std::min(std::max(0, max_corners), n));
  GreedyGrid g(w, h, min_dist, out);
  if (walk_tie_free(cands, g, max_corners)) return true;
  // ---- the tie order matters: hand the survivors and the sort keys of ALL candidates to resolve_ties().  The keys
  // were downloaded speculatively into pinned memory by the device call; take a private copy so that the context
  // can start on the next image.
  if (clk_) clk_->shi_fallbacks++;
  const auto tk0 = Clock::now();
  void* kp = nullptr;
  check(ctx_, sfmx_shi_tomasi_fetch_all_keys(ctx_, n_total, &kp), "shi_tomasi_fetch_all_keys");
  static_assert(sizeof(SortKey) == 16, "SortKey must match the device record {double, u32, u32}");
  if ((int)ties.keys.size() < n_total) ties.keys.resize((size_t)n_total + (size_t)n_total / 8);
  std::memcpy(ties.keys.data(), kp, (size_t)n_total * sizeof(SortKey));
  ties.w = w; ties.h = h; ties.min_dist = min_dist; ties.max_corners = max_corners; ties.n_total = n_total;
  ties.xy.assign(cand_xy_.begin(), cand_xy_.begin() + n);
  ties.s.assign(cand_s_.begin(), cand_s_.begin() + n);
  ties.full.assign(cand_full_.begin(), cand_full_.begin() + n);
  if (clk_) clk_->shi_gpu += since(tk0);
  out.clear();
  return false;
}

bool CornerDetector::resolve_ties(CornerTies& ties, std::vector<V2>& out, StageClock* clk) {
  // tie order from a selective replay of libstdc++'s introsort on the FULL candidate list (introsort_replay.hpp)
  const auto tr0 = Clock::now();
  const int n = (int)ties.xy.size(), nf = ties.n_total;
  SortKey* keys = ties.keys.data();
  std::vector<Cand> cands;
  unpack_survivors(ties.xy.data(), ties.s.data(), n, cands);
  std::vector<Cand> by_score = cands;
  std::sort(by_score.begin(), by_score.end(), [](const Cand& a, const Cand& b) { return a.s > b.s || (a.s == b.s && a.idx < b.idx); });
  // interesting = survivors that share their score with another survivor
  for (size_t a = 0; a + 1 < by_score.size(); a++)
    if (by_score[a].s == by_score[a + 1].s) {
      keys[(size_t)ties.full[(size_t)by_score[a].idx]].mark = 1;
      keys[(size_t)ties.full[(size_t)by_score[a + 1].idx]].mark = 1;
    }
  bool ok = introsort_replay_selective(keys, (size_t)nf);
  if (ok) {
    // positions of the marked elements after the replay (only those are ever compared)
    std::unordered_map<int, int> pos;
    for (int f = 0; f < nf; f++)
      if (keys[(size_t)f].mark) pos.emplace((int)keys[(size_t)f].id, f);
    const std::vector<std::int32_t>& full_of = ties.full;
    std::sort(cands.begin(), cands.end(), [&](const Cand& a, const Cand& b) {
      if (a.s != b.s) return a.s > b.s;
      return pos.at(full_of[(size_t)a.idx]) < pos.at(full_of[(size_t)b.idx]);  // equal scores => both are marked
    });
    out.clear();
    GreedyGrid g(ties.w, ties.h, ties.min_dist, out);
    g.greedy(cands, ties.max_corners);  // the order is now the reference's: plain greedy (T:290-300)
  }
  if (clk) clk->shi_replay += since(tr0);
  return ok;
}

// exact slow path: every candidate, the reference's sort call on the reference's input order
std::vector<V2> CornerDetector::detect_full_sort(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist) {
  int w = 0, h = 0;
  sfmx_pyramid_level_size(pyr, 0, &w, &h);
  const int cap = w * h;
  if ((int)cand_xy_.size() < cap) { cand_xy_.resize((size_t)cap); cand_s_.resize((size_t)cap); cand_full_.resize((size_t)cap); }
  int n = 0;
  double maxv = 0;
  check(ctx_, sfmx_shi_tomasi_candidates(ctx_, pyr, quality, cap, cand_xy_.data(), cand_s_.data(), &n, &maxv), "shi_tomasi_candidates");
  if (clk_) clk_->shi_kernel_us += sfmx_last_kernel_us(ctx_);
  std::vector<Cand> cands((size_t)n);
  for (int i = 0; i < n; i++) cands[(size_t)i] = {(int)(cand_xy_[(size_t)i] & 0xffffu), (int)(cand_xy_[(size_t)i] >> 16), cand_s_[(size_t)i], i};
  std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.s > b.s; });  // T:286
  std::vector<V2> out;
  out.reserve((size_t)std::max(0, max_corners));
  GreedyGrid g(w, h, min_dist, out);
  g.greedy(cands, max_corners);
  return out;
}

std::vector<V2> CornerDetector::detect(sfmx_pyramid* pyr, int max_corners, double quality, int min_dist) {
  std::vector<V2> out;
  if (detect_device(pyr, max_corners, quality, min_dist, out, ties_)) return out;
  if (resolve_ties(ties_, out, clk_)) return out;
  return detect_full_sort(pyr, max_corners, quality, min_dist);  // the replay declined (depth limit): last resort
}

// ------------------------------------------------------------------------------------------ context pool
ContextPool& ContextPool::instance() {
  static ContextPool* p = new ContextPool;  // leaked on purpose, see pipeline.hpp
  return *p;
}
PooledCtx* ContextPool::acquire(int device, int priority, int role) {
  static const bool no_pool = std::getenv("SFMX_NO_CTX_POOL") != nullptr;
  if (!no_pool) {
    std::lock_guard<std::mutex> lk(mu_);
    for (size_t i = 0; i < free_.size(); i++)
      if (free_[i]->device == device && free_[i]->priority == priority && free_[i]->role == role) {
        PooledCtx* pc = free_[i];
        free_.erase(free_.begin() + (long)i);
        return pc;
      }
  }
  auto pc = std::make_unique<PooledCtx>();
  pc->device = device;
  pc->priority = priority;
  pc->role = role;
  check(nullptr, sfmx_ctx_create_prio(device, priority, &pc->ctx), "ctx_create(helper)");
  return pc.release();
}
void ContextPool::release(PooledCtx* pc) {
  if (!pc) return;
  static const bool no_pool = std::getenv("SFMX_NO_CTX_POOL") != nullptr;
  (void)sfmx_sync(pc->ctx);
  (void)sfmx_set_timing(pc->ctx, 0);
  if (no_pool) {
    pc->free_pyramids();
    sfmx_ctx_destroy(pc->ctx);
    delete pc;
    return;
  }
  std::lock_guard<std::mutex> lk(mu_);
  free_.push_back(pc);
}
void ContextPool::clear() {
  std::lock_guard<std::mutex> lk(mu_);
  for (PooledCtx* pc : free_) {
    pc->free_pyramids();
    sfmx_ctx_destroy(pc->ctx);
    delete pc;
  }
  free_.clear();
}
void PooledCtx::free_pyramids() {
  if (ba) sfmx_ba_destroy(ctx, ba);
  ba = nullptr;
  if (pyr) sfmx_pyramid_destroy(ctx, pyr);
  pyr = nullptr;
  for (sfmx_pyramid* p : ring) sfmx_pyramid_destroy(ctx, p);
  ring.clear();
}
const std::vector<sfmx_pyramid*>& PooledCtx::pyramid_ring(int w, int h, int levels, int count) {
  if (!ring.empty() && (rw != w || rh != h || rl != levels || (int)ring.size() != count)) {
    for (sfmx_pyramid* p : ring) sfmx_pyramid_destroy(ctx, p);
    ring.clear();
  }
  while ((int)ring.size() < count) {
    sfmx_pyramid* p = nullptr;
    check(ctx, sfmx_pyramid_create(ctx, w, h, levels, &p), "pyramid_create(helper ring)");
    ring.push_back(p);
  }
  rw = w; rh = h; rl = levels;
  return ring;
}
sfmx_pyramid* PooledCtx::pyramid(int w, int h, int levels) {
  if (pyr && (pw != w || ph != h || pl != levels)) {
    sfmx_pyramid_destroy(ctx, pyr);
    pyr = nullptr;
  }
  if (!pyr) {
    check(ctx, sfmx_pyramid_create(ctx, w, h, levels, &pyr), "pyramid_create(helper)");
    pw = w; ph = h; pl = levels;
  }
  return pyr;
}

// stream priority of a helper context: default, overridable for experiments (-1 high, 0 normal, 1 low)
static int prio_env(const char* name, int dflt) {
  if (const char* e = std::getenv(name)) return std::atoi(e);
  return dflt;
}

// ------------------------------------------------------------------------------------------ prefetcher
CornerPrefetcher::CornerPrefetcher(int device, FrameSource& src, double quality, int min_dist, int workers, bool timing)
    : src_(src), quality_(quality), min_dist_(min_dist) {
  try {
    for (int i = 0; i < std::max(1, workers); ++i) {
      auto w = std::make_unique<Worker>();
      w->pc = ContextPool::instance().acquire(device, prio_env("SFMX_PRIO_PREFETCH", 0), ContextPool::PREFETCH);
      w->ctx = w->pc->ctx;
      workers_.push_back(std::move(w));
      Worker& ww = *workers_.back();
      ww.pyr = ww.pc->pyramid(src.width(), src.height(), 1);
      if (timing) (void)sfmx_set_timing(ww.ctx, 1);
      ww.det = std::make_unique<CornerDetector>(ww.ctx, &ww.clock);
    }
    for (auto& w : workers_) {
      for (auto& b : w->ties) w->free_ties.push_back(&b);
      w->th = std::thread([this, p = w.get()] { run(*p); });
      w->th_resolver = std::thread([this, p = w.get()] { run_resolver(*p); });
    }
  } catch (...) {
    shutdown();
    throw;
  }
}
void CornerPrefetcher::shutdown() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_req_.notify_all();
  for (auto& w : workers_) w->cv_ties.notify_all();
  for (auto& w : workers_) {
    if (w->th.joinable()) w->th.join();
    if (w->th_resolver.joinable()) w->th_resolver.join();
    w->det.reset();
    ContextPool::instance().release(w->pc);
  }
  workers_.clear();
}
CornerPrefetcher::~CornerPrefetcher() { shutdown(); }
void CornerPrefetcher::busy(double& total, double& gpu, double& replay) {
  std::lock_guard<std::mutex> lk(mu_);
  for (auto& w : workers_) { total += w->busy; gpu += w->clock.shi_gpu; replay += w->busy_resolver; }
}
double CornerPrefetcher::kernel_us() {
  double us = 0;
  for (auto& w : workers_) us += w->clock.shi_kernel_us;
  return us;
}
void CornerPrefetcher::grab_profile(StageClock& clk) {
  for (auto& w : workers_) clk.grab_profile(w->ctx);
}
std::uint64_t CornerPrefetcher::replays() {
  std::uint64_t n = 0;
  for (auto& w : workers_) n += w->clock.shi_fallbacks;
  return n;
}
void CornerPrefetcher::request(int frame) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (frame < 0 || frame >= src_.count() || slots_.count(frame)) return;
    slots_.emplace(frame, Slot{});
    queue_.push_back(frame);
  }
  cv_req_.notify_all();
}
void CornerPrefetcher::discard_older_than(int frame) {
  std::lock_guard<std::mutex> lk(mu_);
  for (auto it = slots_.begin(); it != slots_.end();)
    if (it->first < frame && it->second.done) it = slots_.erase(it);
    else ++it;
}
bool CornerPrefetcher::take(int frame, std::vector<V2>& corners) {
  std::unique_lock<std::mutex> lk(mu_);
  auto it = slots_.find(frame);
  if (it == slots_.end()) return false;
  cv_done_.wait(lk, [&] { return slots_.at(frame).done; });
  Slot& s = slots_.at(frame);
  const bool ok = !s.failed;
  if (ok) corners = std::move(s.corners);
  slots_.erase(frame);
  return ok;
}
bool CornerPrefetcher::take_if_done(int frame, std::vector<V2>& corners) {
  std::lock_guard<std::mutex> lk(mu_);
  auto it = slots_.find(frame);
  if (it == slots_.end() || !it->second.done) return false;
  const bool ok = !it->second.failed;
  if (ok) corners = std::move(it->second.corners);
  slots_.erase(it);
  return ok;
}
void CornerPrefetcher::publish(int frame, std::vector<V2>&& seq, bool failed) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = slots_.find(frame);
    if (it != slots_.end()) {
      it->second.corners = std::move(seq);
      it->second.failed = failed;
      it->second.done = true;
    }
  }
  cv_done_.notify_all();
}

// device thread of a worker: image -> device score + fixpoint -> tie-free walk; ties go to the resolver thread
void CornerPrefetcher::run(Worker& w) {
  (void)sfmx_ctx_make_current(w.ctx);  // HIP's current device is per thread
  for (;;) {
    int frame;
    CornerTies* buf = nullptr;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_req_.wait(lk, [&] { return stop_ || (!queue_.empty() && !w.free_ties.empty()); });
      if (stop_) return;
      frame = queue_.front();
      queue_.pop_front();
      buf = w.free_ties.back();
      w.free_ties.pop_back();
    }
    std::vector<V2> seq;
    bool failed = false, final = true;
    const auto tb = Clock::now();
    try {
      src_.load(w.ctx, frame, w.pyr);
      final = w.det->detect_device(w.pyr, 0x3fffffff, quality_, min_dist_, seq, *buf);  // uncapped: every later request is a prefix
    } catch (...) {
      failed = true;  // the main thread recomputes synchronously and reports the error where the reference would
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      w.busy += since(tb);
      if (failed || final) w.free_ties.push_back(buf);
      else w.ties_queue.push_back({frame, buf});
    }
    if (failed || final) publish(frame, std::move(seq), failed);
    else w.cv_ties.notify_one();
  }
}

// resolver thread of a worker: pure host work (introsort replay + greedy pick), no device access
void CornerPrefetcher::run_resolver(Worker& w) {
  for (;;) {
    std::pair<int, CornerTies*> job;
    {
      std::unique_lock<std::mutex> lk(mu_);
      w.cv_ties.wait(lk, [&] { return stop_ || !w.ties_queue.empty(); });
      if (stop_) return;
      job = w.ties_queue.front();
      w.ties_queue.pop_front();
    }
    std::vector<V2> seq;
    bool ok = false;
    const auto tb = Clock::now();
    try {
      ok = CornerDetector::resolve_ties(*job.second, seq, &w.clock_resolver);
    } catch (...) {
      ok = false;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      w.busy_resolver += since(tb);
      w.free_ties.push_back(job.second);
    }
    cv_req_.notify_all();  // a tie buffer is free again
    publish(job.first, std::move(seq), !ok);  // !ok: the replay declined -> the consumer detects synchronously
  }
}

void GpuTracker::reset(FrameSource& src, int fi) {
  const int next = have_prev_ ? (slot_ + 1) % (int)ring_.size() : slot_;
  const auto t0 = Clock::now();
  if (!(preloaded_frame_ == fi && preloaded_slot_ == next)) {
    if (before_load_) before_load_(fi);
    src.load(ctx_, fi, ring_[(size_t)next]);
  }
  preloaded_frame_ = -1;
  if (clk_) clk_->upload += since(t0);
  slot_ = next;
  have_prev_ = true;
  tracks_.clear();
  for (const V2& p : shi_tomasi(ring_[(size_t)slot_], cfg_.max_tracks, cfg_.quality, cfg_.min_distance, fi)) tracks_.push_back({next_id_++, p});
}

void klt_pairs(sfmx_ctx* ctx, const LKConfig& cfg, const sfmx_pyramid* a, const sfmx_pyramid* b, const std::vector<V2>& p0, std::vector<V2>& fwd,
               std::vector<std::uint8_t>& keep, StageClock* clk) {
  const int n = (int)p0.size();
  fwd.resize((size_t)n);
  keep.resize((size_t)n);
  if (n == 0) return;
  static_assert(sizeof(V2) == 16, "V2 must be two packed doubles");
  sfmx_klt_cfg kc{cfg.pyr_levels, cfg.win_radius, cfg.iters, cfg.fb_thresh};
  std::uint64_t steps = 0;
  const auto t0 = Clock::now();
  check(ctx, sfmx_klt_track(ctx, a, b, &p0[0].x, n, &kc, &fwd[0].x, nullptr, keep.data(), &steps), "klt_track");
  if (clk) {
    clk->klt += since(t0);
    clk->klt_kernel_us += sfmx_last_kernel_us(ctx);
    clk->lk_steps += steps;
    clk->tracks_in += (std::uint64_t)n;
    clk->klt_calls++;
  }
}
StepOut GpuTracker::step(FrameSource& src, int fi) {
  if (!have_prev_ || tracks_.empty()) {  // T:341-344
    reset(src, fi);
    return {};
  }
  const int next = (slot_ + 1) % (int)ring_.size();
  sfmx_pyramid* prev = ring_[(size_t)slot_];
  sfmx_pyramid* cur = ring_[(size_t)next];
  const bool ahead = preloaded_frame_ == fi && preloaded_slot_ == next;  // built on the second stream; sfmx_klt_track orders itself behind it
  if (!ahead && before_load_) before_load_(fi);  // the slot's previous tenant must have been released by the geometry lane
  auto t0 = Clock::now();
  if (!ahead) src.load(ctx_, fi, cur);  // pyr1; pyr0 is the cached pyramid of the previous frame (identical to rebuilding it, T:345)
  preloaded_frame_ = -1;
  // the next frame's pyramid, while this frame's KLT runs: into the slot after `cur`, if that slot is free already
  if (may_load_ && ring_.size() >= 3 && fi + 1 < src.count() && may_load_(fi + 1)) {
    const int after = (next + 1) % (int)ring_.size();
    if (src.load_async(ctx_, fi + 1, ring_[(size_t)after], fetch_level_)) { preloaded_frame_ = fi + 1; preloaded_slot_ = after; }
  }
  if (clk_) clk_->upload += since(t0);
  std::vector<V2> p0(tracks_.size()), fwd;
  std::vector<std::uint8_t> keep;
  for (size_t i = 0; i < tracks_.size(); i++) p0[i] = tracks_[i].p;
  klt_pairs(ctx_, cfg_, prev, cur, p0, fwd, keep, clk_);
  StepOut out;
  std::vector<Track> kept;
  kept.reserve(tracks_.size());
  out.prev_pts.reserve(tracks_.size());
  out.cur_pts.reserve(tracks_.size());
  out.ids.reserve(tracks_.size());
  for (size_t i = 0; i < tracks_.size(); i++) {
    if (!keep[i]) continue;  // T:362
    kept.push_back({tracks_[i].id, fwd[i]});
    out.prev_pts.push_back(p0[i]);
    out.cur_pts.push_back(fwd[i]);
    out.ids.push_back(tracks_[i].id);
  }
  slot_ = next;  // prev_ = gray (T:370)
  tracks_ = std::move(kept);
  if ((int)tracks_.size() < cfg_.min_tracks) {  // replenish (T:374-389)
    const int need = cfg_.max_tracks - (int)tracks_.size();
    const auto pts = shi_tomasi(cur, need * 3, cfg_.quality, cfg_.min_distance, fi);
    t0 = Clock::now();
    // distance filter against all live tracks (incl. the ones appended here): existence test, so a
    // grid over track positions gives the same answer as the reference's linear scan.
    const int cell = std::max(1, cfg_.min_distance);
    const int gw = w_ / cell + 3, gh = h_ / cell + 3;  // one guard cell around the image
    grid_head_.assign((size_t)gw * gh, -1);  // per-cell singly linked lists through grid_next_ (no per-cell allocations)
    grid_next_.clear();
    grid_next_.reserve((size_t)cfg_.max_tracks);
    auto cell_of = [&](double v, int lim) -> int {
      if (!(v > -(double)cell && v < (double)(lim + cell))) return -1;  // farther than min_distance from any pixel (or NaN)
      return (int)std::floor(v / cell) + 1;
    };
    auto insert = [&](int ti) {  // ti == grid_next_.size(): tracks are inserted in index order
      const int gx = cell_of(tracks_[(size_t)ti].p.x, w_), gy = cell_of(tracks_[(size_t)ti].p.y, h_);
      if (gx < 0 || gy < 0 || gx >= gw || gy >= gh) { grid_next_.push_back(-2); return; }
      grid_next_.push_back(grid_head_[(size_t)gy * gw + gx]);
      grid_head_[(size_t)gy * gw + gx] = ti;
    };
    for (int i = 0; i < (int)tracks_.size(); i++) insert(i);
    const double md2 = (double)cfg_.min_distance * cfg_.min_distance;
    for (const V2& p : pts) {
      bool ok = true;
      const int cx = (int)p.x / cell + 1, cy = (int)p.y / cell + 1;
      for (int gy = std::max(0, cy - 1); gy <= std::min(gh - 1, cy + 1) && ok; gy++)
        for (int gx = std::max(0, cx - 1); gx <= std::min(gw - 1, cx + 1) && ok; gx++)
          for (int ti = grid_head_[(size_t)gy * gw + gx]; ti >= 0; ti = grid_next_[(size_t)ti]) {
            const double dx = tracks_[(size_t)ti].p.x - p.x, dy = tracks_[(size_t)ti].p.y - p.y;
            if (dx * dx + dy * dy < md2) { ok = false; break; }
          }
      if (!ok) continue;
      tracks_.push_back({next_id_++, p});
      insert((int)tracks_.size() - 1);
      if ((int)tracks_.size() >= cfg_.max_tracks) break;
    }
    if (clk_) clk_->shi += since(t0);
  }
  return out;
}

// ------------------------------------------------------------------------------------------ RANSAC
// find_E_ransac (T:646-761) in two halves.  ransac_local: this rank's share of the hypothesis loop (T:664-677) -- the EXACT
// winner (count, lowest iteration, E, mask) of its contiguous iteration range, no communication.  ransac_merge: the ranks'
// winners -> the call's winner (all-reduce(max) of the packed key, the winner's E as raw bits), then the decomposition
// (T:680-760).  A pipeline runs the first half on whatever lane has the data and the second on the geometry thread, where
// the calls are consumed in program order: every rank issues the collectives of its RANSAC communicator in the same order.
RansacLocal ransac_local(sfmx_ctx* ctx, const Mat3& K, const std::vector<V2>& pi, const std::vector<V2>& pj, int iters, double thr,
                         int min_inliers, StageClock* clk, int rank, int world) {
  RansacLocal out;
  out.thr = thr;
  out.min_inliers = min_inliers;
  if (pi.size() < 8) { out.none = true; return out; }  // T:648
  const auto t0 = Clock::now();
  Mat3 Kinv;
  if (!invert_K(K, Kinv)) throw std::runtime_error("Singular K");  // T:474
  const int n = (int)pi.size();
  out.n = n;
  std::vector<double>& xi = out.xi;
  std::vector<double>& xj = out.xj;
  xi.resize((size_t)2 * n);
  xj.resize((size_t)2 * n);
  for (int i = 0; i < n; i++) {
    const V2 a = norm_point(Kinv, pi[(size_t)i]), b = norm_point(Kinv, pj[(size_t)i]);
    xi[2 * (size_t)i] = a.x; xi[2 * (size_t)i + 1] = a.y;
    xj[2 * (size_t)i] = b.x; xj[2 * (size_t)i + 1] = b.y;
  }
  // the reference draws 8 indices per iteration from mt19937(12345) (T:657-665); pre-draw the stream
  std::vector<std::int32_t> idx8((size_t)8 * std::max(iters, 0));
  {
    // The generator is re-seeded with 12345 on every call, so its raw 32-bit output stream is the same
    // every time; only the range reduction depends on n.  Cache the raw stream; a Lemire rejection
    // (probability < n / 2^32 per draw) consumes one extra raw value exactly as libstdc++ does.
    static thread_local std::vector<std::uint32_t> raw;
    const size_t want = idx8.size() + 64;
    if (raw.size() < want) {
      Mt19937 rng(12345);
      raw.resize(std::max(want, (size_t)8 * 4096 + 64));
      for (auto& v : raw) v = rng.next();
    }
    size_t pos = 0;
    bool exhausted = false;
    const std::uint32_t range = (std::uint32_t)n;
    const std::uint32_t rej = (0u - range) % range;
    for (size_t k = 0; k < idx8.size() && !exhausted; k++) {
      std::uint64_t prod = (std::uint64_t)raw[pos++] * range;
      while ((std::uint32_t)prod < rej) {
        if (pos >= raw.size()) { exhausted = true; break; }
        prod = (std::uint64_t)raw[pos++] * range;
      }
      idx8[k] = (std::int32_t)(prod >> 32);
      if (pos >= raw.size() && k + 1 < idx8.size()) exhausted = true;
    }
    if (exhausted) {  // more rejections than the 64-value margin: regenerate the plain way
      Mt19937 rng(12345);
      for (size_t k = 0; k < idx8.size(); k++) idx8[k] = rng.below(range);
    }
  }
  if (iters <= 0) { out.none = true; return out; }
  if (clk) clk->r_pre += since(t0);
  const auto tg0 = Clock::now();
  // hypothesis sharding: this rank scores iterations [h0, h1) of the common sample stream, with GLOBAL iteration numbers
  int h0 = 0, h1 = iters;
  if (world > 1) sfmx_shard_range(iters, rank, world, &h0, &h1);
  const int hl = h1 - h0;
  std::vector<std::int32_t> counts((size_t)iters, 0), lo((size_t)iters, 0), hi((size_t)iters, 0);
  std::int32_t best_iter = -1, best_count = 0;
  if (hl > 0)
    check(ctx, sfmx_ransac_score_ex(ctx, xi.data(), xj.data(), n, idx8.data() + (size_t)8 * h0, hl, thr, counts.data() + h0, lo.data() + h0,
                                    hi.data() + h0, nullptr, nullptr, &best_iter, &best_count, nullptr),
          "ransac_score");
  if (clk) {
    clk->ransac_kernel_us += sfmx_last_kernel_us(ctx);
    clk->ransac_calls++;
    clk->ransac_points += (std::uint64_t)n;
    clk->r_gpu += since(tg0);
  }
  const auto tv0 = Clock::now();
  // The library reports, per iteration, a count and bounds lo <= reference count <= hi (include/sfmx.h): lo == hi where the
  // hypothesis is the exact host one (repeated-index / ill-conditioned octets) or where no point lies inside the rounding
  // band around thr.  The reference's winner is the LOWEST iteration with the maximal count (strict '>' at T:673); only
  // iterations with hi >= max(lo) can be it.  Of those, the uncertain ones (lo < hi) are re-derived exactly here -- E with
  // the platform libm, mask from that E on the device -- and so is the winner, whose E and mask are what leaves this function.
  int best_lo = 0, best_hi = 0;
  for (int it = h0; it < h1; ++it) { best_lo = std::max(best_lo, lo[(size_t)it]); best_hi = std::max(best_hi, hi[(size_t)it]); }
  if (best_hi > 0 && best_hi >= min_inliers) {
    int& win_iter = out.win_iter;
    int& win_count = out.win_count;
    std::vector<std::uint8_t> mask((size_t)n);
    auto verify = [&](int it) {
      const Mat3 E = eight_point_E(xi.data(), xj.data(), &idx8[(size_t)8 * it]);
      std::int32_t cnt = 0;
      check(ctx, sfmx_sampson_mask(ctx, nullptr, nullptr, n, E.a, thr, mask.data(), &cnt), "sampson_mask");
      if (clk) clk->ransac_verified++;
      if (cnt > win_count || (cnt == win_count && it < win_iter)) {
        win_count = cnt;
        win_iter = it;
        out.winE = E;
        out.win_mask = mask;
      }
      return cnt;
    };
    // pass 1: exact counts of the uncertain candidates; the best certain candidate (count, lowest iteration)
    int cert_iter = -1, cert_count = -1;
    for (int it = h0; it < h1; ++it) {
      if (hi[(size_t)it] < best_lo || hi[(size_t)it] <= 0) continue;
      if (lo[(size_t)it] < hi[(size_t)it]) verify(it);
      else if (counts[(size_t)it] > cert_count) { cert_count = counts[(size_t)it]; cert_iter = it; }
    }
    // pass 2: the best certain candidate beats (or ties earlier than) every verified one?  Then it is this rank's winner and
    // needs its exact E and mask; its exact count must equal the bounded one.
    if (cert_iter >= 0 && (cert_count > win_count || (cert_count == win_count && cert_iter < win_iter))) {
      const int cnt = verify(cert_iter);
      if (cnt != cert_count) {
        // A bound did not hold (never observed), so none of this call's bounds is trusted any more.  Parity first: exact counts
        // of EVERY iteration that has an inlier at all.
        if (clk) clk->ransac_cert_misses++;
        win_iter = -1; win_count = -1;
        for (int it = h0; it < h1; ++it)
          if (hi[(size_t)it] > 0) verify(it);
      }
    }
  }
  if (clk) { clk->r_verify += since(tv0); clk->ransac += since(t0); }
  return out;
}

std::optional<RelPose> ransac_merge(sfmx_ctx* ctx, sfmx_comm* comm, RansacLocal&& loc, StageClock* clk) {
  if (loc.none) return std::nullopt;
  const auto t0 = Clock::now();
  std::optional<RelPose> result;
  const int n = loc.n;
  int win_iter = loc.win_iter, win_count = loc.win_count;
  if (comm && sfmx_comm_world(comm) > 1) {
    // global winner: max count, lowest iteration (T:673) over the ranks' exact local winners; its E travels as raw bits
    // (max over {bits, 0, 0, ...}: exact, signs of zeros included), its mask is recomputed from that E by every other rank
    std::uint64_t key = win_iter >= 0 ? (((std::uint64_t)(std::uint32_t)win_count << 32) | (std::uint64_t)(0x7fffffff - win_iter)) : 0;
    const std::uint64_t mine = key;
    check(ctx, sfmx_comm_allreduce_u64_max(ctx, comm, &key, 1), "allreduce(max) of the winner key");
    std::uint64_t ebits[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (key != 0 && key == mine) std::memcpy(ebits, loc.winE.a, 72);
    if (key != 0) check(ctx, sfmx_comm_allreduce_u64_max(ctx, comm, ebits, 9), "allreduce of the winner's E");  // key is global: every rank skips or none
    if (key == 0) { win_iter = -1; win_count = -1; }
    else if (key != mine) {
      win_count = (int)(key >> 32);
      win_iter = 0x7fffffff - (int)(key & 0xffffffffull);
      std::memcpy(loc.winE.a, ebits, 72);
      if (win_count >= loc.min_inliers) {  // otherwise the gate below drops the result and the mask is never read
        std::int32_t cnt = 0;
        loc.win_mask.resize((size_t)n);
        check(ctx, sfmx_sampson_mask(ctx, loc.xi.data(), loc.xj.data(), n, loc.winE.a, loc.thr, loc.win_mask.data(), &cnt), "sampson_mask");
      }
    }
  }
  if (win_iter >= 0 && win_count >= loc.min_inliers) {  // T:678
    RelPose rp;
    rp.best_iter = win_iter;
    for (int i = 0; i < n; i++)
      if (loc.win_mask[(size_t)i]) rp.inliers.push_back(i);
    decompose_E(loc.winE, loc.xi.data(), loc.xj.data(), rp.inliers, rp.R_ji, rp.t_ji, nullptr,
                [](int n, const std::function<void(int)>& f) { ThreadPool::instance().parallel_for(n, f, 8); });
    result = std::move(rp);
  }
  if (clk) { clk->r_decomp += since(t0); clk->ransac += since(t0); }
  return result;
}

std::optional<RelPose> find_E_ransac_gpu(sfmx_ctx* ctx, const Mat3& K, const std::vector<V2>& pi, const std::vector<V2>& pj, int iters,
                                         double thr, int min_inliers, StageClock* clk, sfmx_comm* comm) {
  return ransac_merge(ctx, comm, ransac_local(ctx, K, pi, pj, iters, thr, min_inliers, clk, sfmx_comm_rank(comm), sfmx_comm_world(comm)), clk);
}

// ------------------------------------------------------------------------------------------ map
int MapState::add(int tid, V3 Xw) {
  const int pid = next_pid++;
  MapPoint mp(pts.get_allocator().arena);
  mp.pid = pid;
  mp.tid = tid;
  mp.Xw = Xw;
  pts.emplace(pid, std::move(mp));
  tid2pid.emplace(tid, pid);
  return pid;
}
void MapState::add_obs(int tid, int kf_id, V2 uv) {
  auto it = tid2pid.find(tid);
  if (it == tid2pid.end()) return;
  pts[it->second].obs.push_back({kf_id, uv});
}

// ------------------------------------------------------------------------------------------ BA
GpuBundleAdjuster::~GpuBundleAdjuster() {
  if (keep_) *keep_ = prob_;  // stays with the pooled context: no hipFree / hipMalloc per run
  else sfmx_ba_destroy(ctx_, prob_);
}

BaJob GpuBundleAdjuster::gather(const Mat3& K, const std::vector<Keyframe>& kfs, const MapState& map, const BAConfig& cfg) {
  BaJob job;
  job.K = K;
  job.cfg = cfg;
  const int N = (int)kfs.size();
  if (N < 2) return job;
  const int w0 = std::max(0, N - cfg.window), W = N - w0;
  if (W < 2) return job;
  std::unordered_map<int, int> kf2local;
  kf2local.reserve((size_t)W);
  for (int li = 0; li < W; ++li) kf2local.emplace(kfs[(size_t)(w0 + li)].kf_id, li);
  // points with >= 2 window observations, in the map's iteration order (T:871-882)
  std::vector<double>& X = job.X;
  std::vector<double>& uv = job.uv;
  std::vector<std::int32_t>& optr = job.optr;
  std::vector<std::int32_t>& oli = job.oli;
  optr.push_back(0);
  int P = 0;
  for (const auto& kv : map.pts) {
    const size_t mark = oli.size();
    for (const auto& ob : kv.second.obs) {
      auto it = kf2local.find(ob.first);
      if (it == kf2local.end()) continue;
      oli.push_back(it->second);
      uv.push_back(ob.second.x);
      uv.push_back(ob.second.y);
    }
    if ((int)(oli.size() - mark) < 2) {
      oli.resize(mark);
      uv.resize(2 * mark);
      continue;
    }
    X.push_back(kv.second.Xw.x); X.push_back(kv.second.Xw.y); X.push_back(kv.second.Xw.z);
    optr.push_back((std::int32_t)oli.size());
    if (++P >= cfg.max_points) break;
  }
  if (P == 0) return job;
  if (W > 64) throw SfmxFailure(SFMX_ERR_UNSUPPORTED, "ba.window > 64 is not supported");
  job.w0 = w0;
  job.W = W;
  job.P = P;
  job.win.resize((size_t)W);
  for (int li = 0; li < W; li++) job.win[(size_t)li] = kfs[(size_t)(w0 + li)].pose;
  job.valid = true;
  return job;
}

void GpuBundleAdjuster::solve(BaJob& job) {
  if (!job.valid) return;
  const auto t0 = Clock::now();
  const int W = job.W, P = job.P, D = 6 * W;
  const Mat3& K = job.K;
  // Multi-GPU run: by default the ELEMENTS of S | b are sharded (sfmx_ba_step_sharded_elements: every rank holds the whole
  // window, results bit-identical to one GPU).  SFMX_BA_SHARD=points selects point sharding instead (tolerance mode: this rank
  // keeps the contiguous range [lo, hi) of the window's points in reference order; with fewer points than ranks every rank
  // runs the whole tiny problem itself, no collective).
  // SFMX_VIRTUAL_WORLD=N (test mode, one rank): the same two entry points form what N ranks would on this one GPU.
  const bool multi = comm_ && sfmx_comm_world(comm_) > 1;
  const char* vw = std::getenv("SFMX_VIRTUAL_WORLD");
  const bool virtual_world = !multi && vw && std::atoi(vw) > 1;
  const char* sm = std::getenv("SFMX_BA_SHARD");
  const bool by_points = sm && std::string(sm) == "points";
  const bool sharded = multi && by_points && P >= sfmx_comm_world(comm_);
  const bool by_elements = (multi || virtual_world) && !by_points;
  const double* Xp = job.X.data();
  const std::int32_t* optr = job.optr.data();
  const std::int32_t* oli = job.oli.data();
  const double* ouv = job.uv.data();
  int Pl = P;
  std::vector<std::int32_t> optr_local;
  if (sharded) {
    int lo = 0, hi = P;
    sfmx_shard_range(P, sfmx_comm_rank(comm_), sfmx_comm_world(comm_), &lo, &hi);
    Pl = hi - lo;
    const std::int32_t o0 = job.optr[(size_t)lo];
    optr_local.resize((size_t)Pl + 1);
    for (int p = 0; p <= Pl; p++) optr_local[(size_t)p] = job.optr[(size_t)(lo + p)] - o0;
    Xp += (size_t)3 * lo;
    optr = optr_local.data();
    oli += o0;
    ouv += (size_t)2 * o0;
  }
  if (!prob_) check(ctx_, sfmx_ba_create(ctx_, W, Pl, Xp, optr, oli, ouv, &prob_), "ba_create");
  else check(ctx_, sfmx_ba_reset(ctx_, prob_, W, Pl, Xp, optr, oli, ouv), "ba_reset");
  std::vector<double> poses((size_t)W * 12), dx((size_t)D);
  if (clk_) clk_->ba_calls++;
  const bool plain = !by_elements && !sharded && !virtual_world;
  struct JobGuard {  // the resident kernel of a job never outlives the job, whatever ends it
    sfmx_ctx* c; sfmx_ba_problem* q; bool on;
    ~JobGuard() { if (on) (void)sfmx_ba_end(c, q); }
  } job_guard{ctx_, prob_, false};
  if (plain) {
    check(ctx_, sfmx_ba_begin(ctx_, prob_, job.cfg.iters, K(0, 0), K(1, 1), K(0, 2), K(1, 2), job.cfg.huber_delta, job.cfg.lambda), "ba_begin");
    job_guard.on = true;
  }
  for (int it = 0; it < job.cfg.iters; ++it) {
    for (int li = 0; li < W; li++) {
      Mat3 R;
      V3 t;
      inv_wc(job.win[(size_t)li], R, t);
      std::memcpy(&poses[(size_t)12 * li], R.a, 72);
      poses[(size_t)12 * li + 9] = t.x; poses[(size_t)12 * li + 10] = t.y; poses[(size_t)12 * li + 11] = t.z;
    }
    const double fx = K(0, 0), fy = K(1, 1), cx = K(0, 2), cy = K(1, 2);
    int rc;
    if (by_elements) rc = sfmx_ba_step_sharded_elements(ctx_, multi ? comm_ : nullptr, prob_, poses.data(), fx, fy, cx, cy, job.cfg.huber_delta, job.cfg.lambda, dx.data());
    else if (sharded || virtual_world) rc = sfmx_ba_step_sharded(ctx_, sharded ? comm_ : nullptr, prob_, poses.data(), fx, fy, cx, cy, job.cfg.huber_delta, job.cfg.lambda, dx.data());
    else rc = sfmx_ba_step(ctx_, prob_, poses.data(), fx, fy, cx, cy, job.cfg.huber_delta, job.cfg.lambda, dx.data());
    if (clk_) { clk_->ba_kernel_us += sfmx_last_kernel_us(ctx_); clk_->ba_iters++; }
    if (rc == SFMX_ERR_SINGULAR) break;  // T:1076-1078: ill-conditioned -> skip the rest of BA
    check(ctx_, rc, "ba_step");
    for (int li = 1; li < W; ++li) {  // T:1081-1095
      const V3 w{dx[(size_t)6 * li], dx[(size_t)6 * li + 1], dx[(size_t)6 * li + 2]};
      const V3 v{dx[(size_t)6 * li + 3], dx[(size_t)6 * li + 4], dx[(size_t)6 * li + 5]};
      Mat3 R;
      V3 t;
      inv_wc(job.win[(size_t)li], R, t);
      const Mat3 R2 = so3_exp(w) * R;
      const V3 t2 = t + v;
      const Mat3 Rcw = transpose(R2);
      job.win[(size_t)li].R = Rcw;
      job.win[(size_t)li].t = -(Rcw * t2);
    }
  }
  if (clk_) clk_->ba += since(t0);
}

void GpuBundleAdjuster::apply(const BaJob& job, std::vector<Keyframe>& kfs) {
  if (!job.valid) return;
  for (int li = 1; li < job.W; ++li) kfs[(size_t)(job.w0 + li)].pose = job.win[(size_t)li];
}

// ------------------------------------------------------------------------------------------ async lane
AsyncLane::AsyncLane(int device, int priority, int role) {
  pc_ = ContextPool::instance().acquire(device, priority, role);
  ctx_ = pc_->ctx;
  th_ = std::thread([this] { run(); });
}
AsyncLane::~AsyncLane() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_task_.notify_all();
  if (th_.joinable()) th_.join();
  ContextPool::instance().release(pc_);
}
// how long a thread polls for a hand-off before it sleeps (SFMX_SPIN_US; 0 = sleep at once)
static std::chrono::microseconds spin_window() {
  // default 0: measured on the 16-core share of a GPU box, polling threads take the cores the other lanes need
  // (300 us: 1 358 / 1 361 keyframes/s against 1 456 / 1 403 without, profiles/r03_ab_spin.txt)
  static const int us = std::getenv("SFMX_SPIN_US") ? std::max(0, std::atoi(std::getenv("SFMX_SPIN_US"))) : 0;
  return std::chrono::microseconds(us);
}
template <class Pred>
static bool spin_until(Pred&& ready, int own_us = -1) {
  const auto win = own_us >= 0 ? std::chrono::microseconds(own_us) : spin_window();
  if (win.count() == 0) return ready();
  const auto t0 = Clock::now();
  for (int i = 0;; ++i) {
    if (ready()) return true;
    __builtin_ia32_pause();
    if ((i & 63) == 63 && Clock::now() - t0 > win) return false;
  }
}
void AsyncLane::submit(std::function<void()> task) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    queue_.push_back(std::move(task));
    submitted_.fetch_add(1, std::memory_order_release);
  }
  cv_task_.notify_one();
}
std::uint64_t AsyncLane::submit_ticket(std::function<void()> task) {
  std::uint64_t t;
  {
    std::lock_guard<std::mutex> lk(mu_);
    queue_.push_back(std::move(task));
    t = submitted_.fetch_add(1, std::memory_order_release) + 1;
  }
  cv_task_.notify_one();
  return t;
}
void AsyncLane::wait_ticket(std::uint64_t ticket) {
  (void)spin_until([&] { return completed_.load(std::memory_order_acquire) >= ticket; }, spin_us_.load(std::memory_order_relaxed));
  std::unique_lock<std::mutex> lk(mu_);
  cv_idle_.wait(lk, [&] { return completed_.load(std::memory_order_acquire) >= ticket; });
  if (error_) {
    std::exception_ptr e = error_;
    error_ = nullptr;
    std::rethrow_exception(e);
  }
}
void AsyncLane::wait() {
  const std::uint64_t want = submitted_.load(std::memory_order_acquire);
  (void)spin_until([&] { return completed_.load(std::memory_order_acquire) >= want; }, spin_us_.load(std::memory_order_relaxed));
  std::unique_lock<std::mutex> lk(mu_);
  cv_idle_.wait(lk, [&] { return queue_.empty() && !busy_; });
  if (error_) {
    std::exception_ptr e = error_;
    error_ = nullptr;
    std::rethrow_exception(e);
  }
}
void AsyncLane::run() {
  (void)sfmx_ctx_make_current(ctx_);
  std::uint64_t taken = 0;
  for (;;) {
    std::function<void()> task;
    (void)spin_until([&] { return submitted_.load(std::memory_order_acquire) > taken; }, spin_us_.load(std::memory_order_relaxed));
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_task_.wait(lk, [&] { return stop_ || !queue_.empty(); });
      if (stop_ && queue_.empty()) return;
      task = std::move(queue_.front());
      queue_.pop_front();
      busy_ = true;
      ++taken;
    }
    const auto tb = Clock::now();
    try {
      task();
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu_);
      if (!error_) error_ = std::current_exception();
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      busy_seconds_ += since(tb);
      busy_ = false;
      completed_.fetch_add(1, std::memory_order_release);
    }
    cv_idle_.notify_all();
  }
}

// T:1131-1197 (assembly on the host, solve through the C ABI).  Up to 3N = kPoseGraphDenseMax unknowns the reference's own
// dense system goes through sfmx_solve_dense (its elimination order, bit-exact).  Beyond that -- the reference needs
// (3N)^2 doubles and O((3N)^3) time there -- the same equations are solved in their structured form H = L (x) I_3 by
// sfmx_posegraph_solve (tolerance mode, ~1e-12 relative to the dense solve).  SFMX_POSEGRAPH_SOLVER=dense|structured
// forces one of them.
static int posegraph_dense_max() {
  if (const char* e = std::getenv("SFMX_POSEGRAPH_SOLVER")) {
    if (std::string(e) == "dense") return 0x7fffffff;
    if (std::string(e) == "structured") return 0;
  }
  return 6400;
}
bool posegraph_optimize_centers(sfmx_ctx* ctx, std::vector<Keyframe>& kfs, const std::vector<PGEdge>& edges) {
  const int N = (int)kfs.size();
  if (N < 2 || edges.empty()) return false;
  const int D = 3 * N;
  const bool dense = D <= posegraph_dense_max();
  std::vector<double> H, g((size_t)D, 0.0), dc((size_t)D, 0.0);
  if (dense) H.assign((size_t)D * D, 0.0);
  // structured form: the distinct lower-triangle entries of L in first-touch order, each summed in edge order (the order
  // the reference's `+=` meets them), the three identical diagonal copies of the dense form folded into one
  std::vector<std::int32_t> ent_ij;
  std::vector<double> ent_v;
  std::unordered_map<std::uint64_t, int> ent_of;
  auto addL = [&](int a, int b, double s) {
    if (dense) {
      for (int d = 0; d < 3; d++) H[(size_t)(3 * a + d) * D + (3 * b + d)] += s;
      return;
    }
    if (b > a) return;  // symmetric: the mirrored call carries the same value
    const std::uint64_t key = ((std::uint64_t)(std::uint32_t)a << 32) | (std::uint32_t)b;
    auto it = ent_of.find(key);
    if (it == ent_of.end()) {
      ent_of.emplace(key, (int)ent_v.size());
      ent_ij.push_back(a);
      ent_ij.push_back(b);
      ent_v.push_back(s);
    } else {
      ent_v[(size_t)it->second] += s;
    }
  };
  for (const PGEdge& e : edges) {
    if (e.i < 0 || e.j < 0 || e.i >= N || e.j >= N) continue;
    const V3 Ci = kfs[(size_t)e.i].pose.t, Cj = kfs[(size_t)e.j].pose.t;
    const V3 dest = Cj - Ci;
    const V3 td = -(transpose(e.R_ji) * e.t_ji);
    const V3 dir = unit(kfs[(size_t)e.i].pose.R * td);
    const double L = std::max(1e-6, norm(dest));
    const V3 dm = L * dir;
    const V3 r = (Cj - Ci) - dm;
    const double w = e.is_loop ? 2.0 : 1.0;
    addL(e.i, e.i, w); addL(e.j, e.j, w); addL(e.i, e.j, -w); addL(e.j, e.i, -w);
    g[(size_t)3 * e.i + 0] += w * (-r.x); g[(size_t)3 * e.i + 1] += w * (-r.y); g[(size_t)3 * e.i + 2] += w * (-r.z);
    g[(size_t)3 * e.j + 0] += w * (r.x);  g[(size_t)3 * e.j + 1] += w * (r.y);  g[(size_t)3 * e.j + 2] += w * (r.z);
  }
  if (dense) {
    for (int d = 0; d < 3; d++) H[(size_t)d * D + d] += 1e9;
  } else {
    addL(0, 0, 1e9);
  }
  for (int d = 0; d < 3; d++) g[(size_t)d] = 0.0;
  int rc;
  if (dense) {
    rc = sfmx_solve_dense(ctx, H.data(), g.data(), D, dc.data());
  } else {
    if (ent_v.empty()) return false;
    // a node no edge touches has an all-zero row: the reference's elimination meets a zero pivot there and gives up
    std::vector<char> touched((size_t)N, 0);
    for (size_t k = 0; k < ent_v.size(); k++) touched[(size_t)ent_ij[2 * k]] = touched[(size_t)ent_ij[2 * k + 1]] = 1;
    for (int i = 0; i < N; i++)
      if (!touched[(size_t)i]) return false;
    rc = sfmx_posegraph_solve(ctx, N, ent_ij.data(), ent_v.data(), (int)ent_v.size(), g.data(), dc.data());
  }
  if (rc == SFMX_ERR_SINGULAR) return false;
  check(ctx, rc, "solve (pose graph)");
  for (int i = 1; i < N; i++) {
    kfs[(size_t)i].pose.t.x += dc[(size_t)3 * i];
    kfs[(size_t)i].pose.t.y += dc[(size_t)3 * i + 1];
    kfs[(size_t)i].pose.t.z += dc[(size_t)3 * i + 2];
  }
  return true;
}

// ------------------------------------------------------------------------------------------ main loop
namespace {

// global_desc_32 (T:1100-1122): the repeated downsample2 chain IS the device pyramid, so only the
// first level with w<=32 && h<=32 is downloaded.
int desc_level(int w, int h) {
  int l = 0;
  while (w > 32 || h > 32) { w /= 2; h /= 2; l++; }
  return l;
}
std::vector<float> global_desc_32(sfmx_ctx* ctx, sfmx_pyramid* pyr, int level) {
  int dw = 0, dh = 0;
  sfmx_pyramid_level_size(pyr, level, &dw, &dh);
  std::vector<std::uint8_t> d((size_t)std::max(1, dw * dh));
  const std::uint8_t* fetched = nullptr;  // the level came to the host with an asynchronous build of this pyramid?
  check(ctx, sfmx_pyramid_fetched_level(ctx, pyr, level, &fetched), "pyramid_fetched_level");
  if (fetched) std::memcpy(d.data(), fetched, (size_t)dw * dh);
  else check(ctx, sfmx_pyramid_download_level(ctx, pyr, level, d.data()), "pyramid_download_level");
  std::vector<float> v;
  v.reserve(1024);
  double mean = 0.0;
  for (int y = 0; y < 32; y++)
    for (int x = 0; x < 32; x++) {
      const int sx = std::min(dw - 1, (int)std::round((double)x * (dw - 1) / 31.0));
      const int sy = std::min(dh - 1, (int)std::round((double)y * (dh - 1) / 31.0));
      const float val = (float)d[(size_t)sy * dw + sx];
      v.push_back(val);
      mean += val;
    }
  mean /= (32.0 * 32.0);
  double n2 = 0.0;
  for (float& x : v) { x = (float)(x - (float)mean); n2 += (double)x * (double)x; }
  const double inv = 1.0 / std::sqrt(n2 + 1e-12);
  for (float& x : v) x = (float)(x * inv);
  return v;
}
float dot_desc(const std::vector<float>& a, const std::vector<float>& b) {
  float s = 0.0f;
  const size_t n = std::min(a.size(), b.size());
  for (size_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

}  // namespace

// ------------------------------------------------------------------------------------------ tracker lane
void StageClock::grab_profile(sfmx_ctx* ctx) {
  double us[kKernels] = {};
  std::uint64_t calls[kKernels] = {};
  const int n = sfmx_kernel_profile(ctx, 1, kKernels, us, calls);
  for (int i = 0; i < n && i < kKernels; i++) { kernel_us[i] += us[i]; kernel_calls[i] += calls[i]; }
}
void StageClock::add(const StageClock& o) {
  for (int i = 0; i < kKernels; i++) { kernel_us[i] += o.kernel_us[i]; kernel_calls[i] += o.kernel_calls[i]; }
  klt += o.klt; shi += o.shi; ransac += o.ransac; ba += o.ba; upload += o.upload; host += o.host; total += o.total; shi_gpu += o.shi_gpu;
  shi_replay += o.shi_replay; desc += o.desc; bookkeeping += o.bookkeeping;
  r_pre += o.r_pre; r_gpu += o.r_gpu; r_verify += o.r_verify; r_decomp += o.r_decomp; tri_iter += o.tri_iter; tri_solve += o.tri_solve;
  tri_insert += o.tri_insert;
  klt_kernel_us += o.klt_kernel_us; ransac_kernel_us += o.ransac_kernel_us; ba_kernel_us += o.ba_kernel_us; shi_kernel_us += o.shi_kernel_us;
  lk_steps += o.lk_steps; tracks_in += o.tracks_in; ransac_calls += o.ransac_calls; ransac_points += o.ransac_points; ba_calls += o.ba_calls;
  ba_iters += o.ba_iters; klt_calls += o.klt_calls; ransac_verified += o.ransac_verified; ransac_cert_misses += o.ransac_cert_misses; shi_fallbacks += o.shi_fallbacks;
  shi_calls += o.shi_calls; shi_memo_hits += o.shi_memo_hits; shi_prefetched += o.shi_prefetched;
  shi_wait += o.shi_wait; setup += o.setup;
  pf_busy += o.pf_busy; pf_gpu += o.pf_gpu; pf_replay += o.pf_replay; lane_b_busy += o.lane_b_busy; lane_c_busy += o.lane_c_busy; lane_e_busy += o.lane_e_busy;
  lane_a_busy += o.lane_a_busy;
  join_wait += o.join_wait; ba_gather += o.ba_gather; m_step += o.m_step; m_ransac += o.m_ransac; m_kf += o.m_kf; feed_wait += o.feed_wait;
}

FrameFeeder::FrameFeeder(sfmx_ctx* caller_ctx, FrameSource& src, const LKConfig& cfg, int extra_levels, int desc_level, int n_frames,
                         bool threaded, CornerPrefetcher* prefetch, int prefetch_depth, StageClock* clk,
                         std::function<void(FramePacket&)> on_packet)
    : src_(src), desc_level_(desc_level), n_frames_(n_frames), prefetch_depth_(prefetch_depth), prefetch_(prefetch),
      on_packet_(std::move(on_packet)), ctx_(caller_ctx), clk_(clk) {
  if (threaded) {
    pc_ = ContextPool::instance().acquire(sfmx_ctx_device(caller_ctx), prio_env("SFMX_PRIO_TRACKER", 0), ContextPool::TRACKER);
    ctx_ = pc_->ctx;
    if (sfmx_get_timing(caller_ctx)) (void)sfmx_set_timing(ctx_, 1);
    clk_ = &lane_clk_;
    ring_ = 6;
  }
  try {
    std::function<void(int)> hook;
    if (threaded)
      hook = [this](int fi) {  // frame fi is about to overwrite the pyramid of frame fi - ring
        (void)spin_until([&] { return released_a_.load(std::memory_order_acquire) >= fi - ring_; });
        std::unique_lock<std::mutex> lk(mu_);
        cv_rel_.wait(lk, [&] { return stop_ || released_ >= fi - ring_; });
        if (stop_) throw SfmxFailure(SFMX_ERR_INVALID, "tracker lane stopped");
      };
    const std::vector<sfmx_pyramid*>* borrowed = nullptr;
    if (pc_) borrowed = &pc_->pyramid_ring(src.width(), src.height(), std::max(cfg.pyr_levels, extra_levels), ring_);
    tracker_ = std::make_unique<GpuTracker>(ctx_, cfg, src.width(), src.height(), extra_levels, clk_, ring_, hook, borrowed);
    tracker_->set_prefetcher(prefetch_);
    if (threaded && !std::getenv("SFMX_NO_PRELOAD"))
      tracker_->enable_preload([this](int fi) {
        std::lock_guard<std::mutex> lk(mu_);
        return !stop_ && released_ >= fi - ring_;
      }, desc_level_);
    if (prefetch_)  // the corners of the first frames are on their way before the tracker asks for them (T:330 at frame 0)
      for (int a = 0; a <= prefetch_depth_ && a < n_frames_; ++a) prefetch_->request(a);
    if (threaded) th_ = std::thread([this] { run(); });
  } catch (...) {
    tracker_.reset();
    if (pc_) ContextPool::instance().release(pc_);
    throw;
  }
}
FrameFeeder::~FrameFeeder() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    stop_ = true;
  }
  cv_rel_.notify_all();
  cv_pkt_.notify_all();
  if (th_.joinable()) th_.join();
  tracker_.reset();  // its pyramids live on the lane's context
  if (pc_) ContextPool::instance().release(pc_);
}
FramePacket FrameFeeder::produce(int fi) {
  if (prefetch_) {
    prefetch_->discard_older_than(fi);  // results nobody asked for (no replenish on that frame)
    for (int a = 1; a <= prefetch_depth_; ++a)
      if (fi + a < n_frames_) prefetch_->request(fi + a);
  }
  FramePacket p;
  p.fi = fi;
  const auto t0 = Clock::now();
  p.step = tracker_->step(src_, fi);
  p.tracks = tracker_->tracks();
  p.pyr = tracker_->current();
  p.corners = tracker_->take_memo(fi);
  if (!p.corners && prefetch_) {  // no replenish on this frame: the sequence computed ahead anyway still serves the loop
    std::vector<V2> seq;          // closure's corner request on this image (T:1841), if it has arrived by now
    if (prefetch_->take_if_done(fi, seq))
      p.corners = std::make_shared<const CornerMemo>(CornerMemo{prefetch_->quality(), prefetch_->min_dist(), 0x7fffffff, true, std::move(seq)});
  }
  clk_->m_step += since(t0);
  const auto td = Clock::now();
  if (on_packet_) on_packet_(p);  // starts the frame->frame RANSAC on lane A before the descriptor download below
  p.desc = global_desc_32(ctx_, tracker_->current(), desc_level_);
  clk_->desc += since(td);
  return p;
}
void FrameFeeder::run() {
  (void)sfmx_ctx_make_current(ctx_);
  try {
    for (int fi = 0; fi < n_frames_; ++fi) {
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (stop_) break;
      }
      FramePacket p = produce(fi);
      {
        std::lock_guard<std::mutex> lk(mu_);
        queue_.push_back(std::move(p));
        produced_.fetch_add(1, std::memory_order_release);
      }
      cv_pkt_.notify_one();
    }
  } catch (...) {
    std::lock_guard<std::mutex> lk(mu_);
    error_ = std::current_exception();
  }
  {
    std::lock_guard<std::mutex> lk(mu_);
    done_ = true;
  }
  cv_pkt_.notify_all();
}
FramePacket FrameFeeder::next() {
  if (!pc_) return produce(next_frame_++);
  (void)spin_until([&] { return produced_.load(std::memory_order_acquire) > consumed_; });
  std::unique_lock<std::mutex> lk(mu_);
  cv_pkt_.wait(lk, [&] { return !queue_.empty() || done_; });
  if (queue_.empty()) {
    if (error_) std::rethrow_exception(error_);
    throw SfmxFailure(SFMX_ERR_INVALID, "tracker lane ended early");
  }
  FramePacket p = std::move(queue_.front());
  queue_.pop_front();
  ++consumed_;
  return p;
}
void FrameFeeder::finish() {
  if (th_.joinable()) th_.join();
}
void FrameFeeder::release_upto(int frame) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    if (frame <= released_) return;
    released_ = frame;
    released_a_.store(frame, std::memory_order_release);
  }
  cv_rel_.notify_all();
}

void run_pipeline(sfmx_ctx* ctx, FrameSource& src, const std::vector<FrameMeta>& meta, const Mat3& K, const PipelineConfig& cfg,
                  PipelineResult& out, void (*echo)(const std::string&)) {
  const auto t_all = Clock::now();
  struct PhaseMark {  // SFMX_TRACE_PHASES=1: prints when the objects declared AFTER it have been destroyed
    const char* what; Clock::time_point t0;
    ~PhaseMark() { if (std::getenv("SFMX_TRACE_PHASES")) std::fprintf(stderr, "phase %-22s %8.3f ms\n", what, std::chrono::duration<double>(Clock::now() - t0).count() * 1e3); }
  };
  StageClock& clk = out.clock;
  const int w = src.width(), h = src.height();
  const int dlevel = desc_level(w, h);
  if (dlevel + 1 > 8 || cfg.klt.pyr_levels > 8) throw SfmxFailure(SFMX_ERR_UNSUPPORTED, "image too large for an 8-level pyramid");
  // one worker keeps frame f+1 in flight; more (SFMX_PREFETCH_WORKERS) remove the residual wait but the extra
  // contexts slow the other lanes down by more than that on one GPU (measured: 1 -> 552, 2 -> 498, 3 -> 510 kf/s)
  // Stream creation order decides which hardware queue a lane's stream lands on (csrc/hip/ctx.hip), and the pool hands
  // every lane the same context again on later runs.  The helper contexts are therefore created up front, in a fixed
  // order, the first time a device is used: T(racker) B P(refetch) C -- all 24 orders were measured on MI355X with
  // GPU_MAX_HW_QUEUES=8 (705-877 keyframes/s; the BA lane must not end up on the geometry lane's queue, which is where the
  // fourth context goes).  SFMX_LANE_ORDER overrides it for experiments.
  {
    static std::mutex once_mu;
    static std::vector<int> warmed_devices;
    std::lock_guard<std::mutex> lk(once_mu);
    const int dev = sfmx_ctx_device(ctx);
    if (std::find(warmed_devices.begin(), warmed_devices.end(), dev) == warmed_devices.end() && !std::getenv("SFMX_NO_ASYNC") &&
        !std::getenv("SFMX_NO_CTX_POOL")) {
      warmed_devices.push_back(dev);
      const char* order = std::getenv("SFMX_LANE_ORDER");
      if (!order) order = "TBPCAE";
      std::vector<PooledCtx*> made;
      for (const char* c = order; *c; ++c) {
        const int role = *c == 'P' ? ContextPool::PREFETCH : *c == 'T' ? ContextPool::TRACKER : *c == 'B' ? ContextPool::LANE_B
                         : *c == 'C' ? ContextPool::LANE_C : *c == 'A' ? ContextPool::LANE_A : *c == 'E' ? ContextPool::LANE_E
                         : *c == 'a' ? ContextPool::LANE_A2 : 0;  // (two 'P' = both prefetch workers' contexts)
        if (role) made.push_back(ContextPool::instance().acquire(dev, 0, role));
      }
      for (PooledCtx* pc : made) ContextPool::instance().release(pc);
    }
  }
  // one worker's 0.9 ms per frame was the tracker lane's bound at 640x480 (it waited 26 of 52 ms for corners): two there.  At
  // 1920x1080 a frame has ~600 k candidates and the tie-order replay on the worker's resolver thread takes milliseconds: with two
  // workers the tracker waited 0.37 s of a 0.41 s pass for corners (C5 prefix, 141 frames/s), with four 0.19 of 0.25 s (248 frames/s)
  int prefetch_workers = (size_t)w * (size_t)h >= (size_t)1000000 ? 6 : 2;
  if (const char* e = std::getenv("SFMX_PREFETCH_WORKERS")) prefetch_workers = std::min(8, std::max(1, std::atoi(e)));
  PhaseMark pm_prefetch{"~prefetch.. done", t_all};
  std::unique_ptr<CornerPrefetcher> prefetch;
  if (!std::getenv("SFMX_NO_PREFETCH") && std::min(cfg.frames, src.count()) > 1 && cfg.klt.min_distance >= 1 && cfg.klt.min_distance <= 16) {
    prefetch = std::make_unique<CornerPrefetcher>(sfmx_ctx_device(ctx), src, cfg.klt.quality, cfg.klt.min_distance, prefetch_workers, sfmx_get_timing(ctx) != 0);
  }
  // Tracker lane (FrameFeeder): SFMX_NO_ASYNC / SFMX_NO_TRACK_LANE run KLTTracker::step inline on the caller's context.
  const int n_frames = std::min(cfg.frames, src.count());
  const bool track_lane = !std::getenv("SFMX_NO_ASYNC") && !std::getenv("SFMX_NO_TRACK_LANE") && n_frames > 1;
  // Lane A: the frame->frame find_E_ransac (T:1739) is a pure function of the tracker's StepOut (RNG seeded inside, T:657), so
  // it starts as soon as the tracker lane has produced the packet, on a context of its own; the geometry lane picks the result
  // up with the packet (SFMX_NO_RANSAC_LANE=1: computed by the geometry lane itself, as before).
  PhaseMark pm_lane_a{"~lane_a.. done", t_all};
  StageClock lane_a_clk, lane_a2_clk;  // declared before the lanes: their tasks may still run while the lanes are torn down
  std::unique_ptr<AsyncLane> lane_a, lane_a2;  // lane_a2 (SFMX_RANSAC_LANES=2): odd frames, so that a call has two frame times
  if (track_lane && !std::getenv("SFMX_NO_RANSAC_LANE")) {
    lane_a = std::make_unique<AsyncLane>(sfmx_ctx_device(ctx), prio_env("SFMX_PRIO_LANE_A", 0), ContextPool::LANE_A);
    // a second lane for the odd frames: a call (kernels + exact host hypotheses + decomposition) takes ~490 us of lane time at the
    // reference's 2 200 tracks, more than the geometry thread needs per frame since it stopped waiting for the loop verdict first
    // (29.9 -> 29.0 ms per 47-frame pass, profiles/r03_ab_inproc_rl.txt); with 5 000 tracks per frame (C3) lane A was busy 1.23 s
    // of a 2.2 s pass: 437 -> 543 frames/s with two lanes (SFMX_RANSAC_LANES=1|2 overrides)
    int ransac_lanes = 2;
    if (const char* e = std::getenv("SFMX_RANSAC_LANES")) ransac_lanes = std::atoi(e);
    if (ransac_lanes >= 2) lane_a2 = std::make_unique<AsyncLane>(sfmx_ctx_device(ctx), prio_env("SFMX_PRIO_LANE_A", 0), ContextPool::LANE_A2);
  }
  if (lane_a && sfmx_get_timing(ctx)) (void)sfmx_set_timing(lane_a->ctx(), 1);
  if (lane_a2 && sfmx_get_timing(ctx)) (void)sfmx_set_timing(lane_a2->ctx(), 1);
  // multi-GPU run: a lane computes this rank's half of a RANSAC call (ransac_local), the geometry thread merges it with the
  // other ranks' where it consumes the result -- the one place whose order is the same on every rank
  sfmx_comm* const rcomm = cfg.comm_ransac;
  const int r_rank = sfmx_comm_rank(rcomm), r_world = sfmx_comm_world(rcomm);
  const bool r_sharded = r_world > 1;
  auto ransac_ahead = [r_rank, r_world, r_sharded](sfmx_ctx* c, const Mat3& Km, const std::vector<V2>& a, const std::vector<V2>& b, int iters,
                                                   double thr, int min_inl, StageClock* ck) {
    auto out = std::make_shared<RansacAhead>();
    out->local = ransac_local(c, Km, a, b, iters, thr, min_inl, ck, r_rank, r_world);
    out->merged = !r_sharded;
    if (out->merged) out->rel = ransac_merge(c, nullptr, std::move(out->local), ck);
    return out;
  };
  auto ransac_finish = [&](RansacAhead& ra) -> std::optional<RelPose> {  // geometry thread only
    if (!ra.merged) {
      ra.rel = ransac_merge(ctx, rcomm, std::move(ra.local), &clk);
      ra.merged = true;
    }
    return ra.rel;
  };
  std::function<void(FramePacket&)> on_packet;
  if (lane_a)
    on_packet = [&lane_a, &lane_a2, &lane_a_clk, &lane_a2_clk, K, ransac_ahead](FramePacket& p) {
      if (p.step.prev_pts.empty()) return;
      auto prom = std::make_shared<std::promise<std::shared_ptr<RansacAhead>>>();
      p.rel = prom->get_future().share();
      auto pi = std::make_shared<const std::vector<V2>>(p.step.prev_pts);
      auto pj = std::make_shared<const std::vector<V2>>(p.step.cur_pts);
      const bool second = lane_a2 && (p.fi & 1);
      AsyncLane* la = second ? lane_a2.get() : lane_a.get();
      StageClock* ck = second ? &lane_a2_clk : &lane_a_clk;
      la->submit([prom, pi, pj, la, ck, K, ransac_ahead]() {
        try {
          prom->set_value(ransac_ahead(la->ctx(), K, *pi, *pj, 2500, 1e-3, 60, ck));
        } catch (...) {
          prom->set_exception(std::current_exception());
        }
      });
    };
  PhaseMark pm_feeder{"~feeder.. done", t_all};
  // How many frames ahead of the tracker the corner requests run.  A result takes ~1.5 ms (device fixpoint + tie-order replay on
  // the resolver thread) and the tracker needs one every ~0.65 ms (it replenishes on nearly every frame): with workers + 1 frames
  // of lead it waited ~6 ms per 47-frame pass for corners; a result is a few KB, so a deeper queue costs nothing.
  int prefetch_depth = prefetch_workers + 4;
  if (const char* e = std::getenv("SFMX_PREFETCH_DEPTH")) prefetch_depth = std::min(64, std::max(1, std::atoi(e)));
  FrameFeeder feeder(ctx, src, cfg.klt, dlevel + 1, dlevel, n_frames, track_lane, prefetch.get(), prefetch_depth, &clk, on_packet);
  CornerDetector geo_det(ctx, &clk);                                  // loop closure: corners of old keyframe images ...
  std::unordered_map<int, std::shared_ptr<const CornerMemo>> kf_corners;  // ... unless their sequence is already known
  // only the first `loop_corners` corners of a keyframe image are ever asked for again (T:1838-1841): keep that prefix
  const int loop_corners = 1200;
  auto keep_corners = [&](int frame, const std::shared_ptr<const CornerMemo>& m) {
    if (!m) return;
    if ((int)m->corners.size() <= loop_corners) { kf_corners[frame] = m; return; }
    CornerMemo cut{m->quality, m->min_dist, std::min(m->cap, loop_corners), false, m->prefix(loop_corners)};
    kf_corners[frame] = std::make_shared<const CornerMemo>(std::move(cut));
  };
  // Lane B: the local BA of keyframe k (T:1820) does not feed frame k+1's tracking or frame->frame RANSAC, so its device
  // iterations run on a second context while this thread goes on; it is joined right before the next keyframe's
  // triangulation reads the refined poses, and before any pose-graph use.
  // Lane C: the keyframe->keyframe RANSAC (T:1793; its edge only feeds the pose graph / CSV) and the loop-closure
  // verification of keyframe k (KLT old-keyframe -> new-keyframe + RANSAC, T:1834-1858), whose verdict is only consumed
  // -- pose graph + second BA, T:1859-1863 -- before the next keyframe is built.
  // Lane E: the keyframe->keyframe RANSAC (T:1793).  Its edge feeds nothing but the pose graph and the CSV, so it is not
  // joined per keyframe at all: the results are appended to `edges` in keyframe order when a loop closure needs them
  // (finish_loop) and at the end.  SFMX_NO_EDGE_LANE=1 (and a sharded run without a fourth communicator) keeps these on
  // lane C, joined at every keyframe.
  // SFMX_NO_ASYNC=1 runs everything on this thread.
  const bool use_lane = !std::getenv("SFMX_NO_ASYNC");
  PhaseMark pm_lanes{"~lanes B C E.. done", t_all};
  StageClock lane_clk, lane_c_clk, lane_e_clk;
  std::unique_ptr<AsyncLane> lane, lane_c, lane_e;
  if (use_lane) {
    lane = std::make_unique<AsyncLane>(sfmx_ctx_device(ctx), prio_env("SFMX_PRIO_LANE_B", 0), ContextPool::LANE_B);
    // SFMX_SPIN_B_US: polling window of the ONE hand-off pair on the pass's critical chain -- the geometry thread waiting for BA(k)
    // and lane B waiting for the next job (twice per keyframe; every other lane keeps sleeping at once)
    if (const char* sb = std::getenv("SFMX_SPIN_B_US")) lane->set_spin_us(std::max(0, std::atoi(sb)));
    lane_c = std::make_unique<AsyncLane>(sfmx_ctx_device(ctx), prio_env("SFMX_PRIO_LANE_C", 0), ContextPool::LANE_C);
    if (!std::getenv("SFMX_NO_EDGE_LANE"))
      lane_e = std::make_unique<AsyncLane>(sfmx_ctx_device(ctx), prio_env("SFMX_PRIO_LANE_E", 0), ContextPool::LANE_E);
  }
  if (sfmx_get_timing(ctx)) {  // per-kernel event timing is inherited by the helper contexts
    if (lane) (void)sfmx_set_timing(lane->ctx(), 1);
    if (lane_c) (void)sfmx_set_timing(lane_c->ctx(), 1);
    if (lane_e) (void)sfmx_set_timing(lane_e->ctx(), 1);
  }
  sfmx_ctx* bctx = lane ? lane->ctx() : ctx;
  StageClock* bclk = lane ? &lane_clk : &clk;
  sfmx_ctx* cctx = lane_c ? lane_c->ctx() : ctx;
  StageClock* cclk = lane_c ? &lane_c_clk : &clk;
  AsyncLane* edge_lane = lane_e ? lane_e.get() : lane_c.get();  // where the keyframe->keyframe RANSAC runs
  // the loop verification of a keyframe goes to lane C at the top of the keyframe's block (it needs a lane C of its own: the join
  // then waits for that one task, not for the lane); SFMX_VERIFY_LATE=1: where the reference has it, after the BA submit (A/B, tests)
  const bool verify_early = lane_c && lane_e && !std::getenv("SFMX_VERIFY_LATE");
  sfmx_ctx* ectx = lane_e ? lane_e->ctx() : cctx;
  StageClock* eclk = lane_e ? &lane_e_clk : cclk;
  GpuBundleAdjuster ba(bctx, bclk, cfg.comm_ba, lane ? &lane->pooled()->ba : nullptr);
  sfmx_pyramid* old_pyr_c = nullptr;  // lane C's copy of the old keyframe image
  struct PyrGuardC { sfmx_ctx* c; sfmx_pyramid** p; ~PyrGuardC() { if (*p) sfmx_pyramid_destroy(c, *p); } } guard_c{cctx, &old_pyr_c};
  struct PendingEdge { int i, j; std::shared_ptr<RansacAhead> ra; };
  std::deque<PendingEdge> pending_edges;
  BaJob pending_ba;
  // loop-closure verification of a keyframe (lane C): `cell` receives its RANSAC result, `ticket` is the lane task to wait for.
  // queued_loop: submitted at the START of a keyframe block (SFMX_VERIFY_LATE unset), promoted to pending_loop where the
  // reference runs the detection (after the keyframe's BA, T:1822) -- the verdict is consumed at the next keyframe either way
  struct PendingLoop {
    bool active = false;
    int frame = -1, old_kf = -1, new_kf = -1;
    std::shared_ptr<std::shared_ptr<RansacAhead>> cell;
    std::optional<RelPose> rel;
    std::uint64_t ticket = 0;
  } pending_loop, queued_loop;
  struct LaneGuard {  // declared after everything the lanes' tasks reference: drained first when unwinding
    AsyncLane *l, *m, *e;
    ~LaneGuard() {
      if (l) { try { l->wait(); } catch (...) {} }
      if (m) { try { m->wait(); } catch (...) {} }
      if (e) { try { e->wait(); } catch (...) {} }
    }
  } lane_guard{lane.get(), lane_c.get(), lane_e.get()};
  sfmx_pyramid* old_pyr = nullptr;  // loop-closure verification image (T:1834)
  struct Guard { sfmx_ctx* c; sfmx_pyramid** p; ~Guard() { if (*p) sfmx_pyramid_destroy(c, *p); } } guard{ctx, &old_pyr};

  Pose cur;
  std::vector<Keyframe>& kfs = out.kfs;
  MapState& map = out.map;
  std::vector<PGEdge>& edges = out.edges;
  // Joining is split so that the wait for BA(k) can be pushed as late as the data dependence allows:
  //   join_c: lane C is done; returns true if the loop closure of keyframe k was accepted
  //   flush_edges: the keyframe->keyframe edges found so far, appended in keyframe order (T:1798 comes before T:1859)
  //   join_b: lane B is done -> write BA(k)'s poses back
  //   finish_loop: loop edge + pose graph + second BA (T:1859-1863), needs all three
  std::optional<PendingLoop> accepted_loop;
  auto flush_edges = [&]() {
    const auto tj = Clock::now();
    if (edge_lane) edge_lane->wait();
    clk.join_wait += since(tj);
    for (PendingEdge& pe : pending_edges) {  // keyframe order = the order every rank merges them in
      if (!pe.ra) continue;
      const std::optional<RelPose> rel = ransac_finish(*pe.ra);
      if (rel) edges.push_back(PGEdge{pe.i, pe.j, rel->R_ji, rel->t_ji, (int)rel->inliers.size(), false});
    }
    pending_edges.clear();
  };
  auto join_c = [&]() -> bool {
    const auto tj = Clock::now();
    if (lane_c) {
      // with the verification of the NEXT keyframe possibly queued behind it, only the pending one is waited for
      if (pending_loop.active && pending_loop.ticket) lane_c->wait_ticket(pending_loop.ticket);
      else if (!queued_loop.active) lane_c->wait();
    }
    clk.join_wait += since(tj);
    if (pending_loop.active) {  // verdict of the loop-closure verification of the last keyframe (T:1858)
      PendingLoop pl = pending_loop;
      pending_loop = PendingLoop{};
      if (pl.cell && *pl.cell) pl.rel = ransac_finish(**pl.cell);
      if (pl.rel && (int)pl.rel->inliers.size() >= 100) accepted_loop = pl;
    }
    return accepted_loop.has_value();
  };
  double t_join_b = 0;  // (SFMX_TRACE_PHASES: lane B's share of join_wait)
  auto join_b = [&]() {
    const auto tj = Clock::now();
    if (lane) lane->wait();
    clk.join_wait += since(tj);
    t_join_b += since(tj);
    GpuBundleAdjuster::apply(pending_ba, kfs);
    pending_ba = BaJob{};
  };
  auto finish_loop = [&]() {
    const PendingLoop pl = *accepted_loop;
    accepted_loop.reset();
    flush_edges();
    edges.push_back(PGEdge{pl.old_kf, pl.new_kf, pl.rel->R_ji, pl.rel->t_ji, (int)pl.rel->inliers.size(), true});
    (void)posegraph_optimize_centers(ctx, kfs, edges);
    ba.run(K, kfs, map, cfg.ba);  // the lanes are idle here: lane B's context is used from this thread
  };
  auto join_lane = [&]() {
    const bool looped = join_c();
    join_b();
    if (looped) finish_loop();
    if (!lane_e) flush_edges();  // without lane E the edges share lane C, which has just been joined
  };
  PhaseMark pm_maps{"~maps.. done", t_all};
  std::vector<std::vector<float>> kf_desc;
  Arena* arena = out.arena.get();
  // mapped: the track already has a map point (== map.has(tid), kept in the node the walk below touches anyway)
  // (the observation lists live in the arena too: ~15 000 heap vectors cost 1.7 ms to free at the end of a 47-frame run)
  using ObsList = std::vector<std::pair<int, V2>, ArenaAlloc<std::pair<int, V2>>>;
  struct TrackHist {
    ObsList obs;
    bool mapped = false;
    explicit TrackHist(Arena* a) : obs(ArenaAlloc<std::pair<int, V2>>(a)) {}
  };
  ArenaMap<int, TrackHist> track_hist(0, std::hash<int>(), std::equal_to<int>(), ArenaAlloc<std::pair<const int, TrackHist>>(arena));
  int last_kf_frame = -999999;
  const int frames = cfg.frames;
  std::ostringstream so;
  auto emit = [&](int fi) {
    std::ostringstream line;
    line << "frame " << (fi + 1) << "/" << frames << " | keyframes=" << kfs.size() << " | map_points=" << map.pts.size() << "\n";
    so << line.str();
    if (echo) echo(line.str());
  };

  clk.setup = since(t_all);
  static const bool trace_phases = std::getenv("SFMX_TRACE_PHASES") != nullptr;
  auto phase = [&](const char* what) { if (trace_phases) std::fprintf(stderr, "phase %-22s %8.3f ms\n", what, since(t_all) * 1e3); };
  phase("setup done");
  double t_par = 0, t_emit = 0, t_rel = 0, t_first = 0, t_loopjoin = 0;
  for (int fi = 0; fi < std::min(frames, src.count()); ++fi) {
    const auto tlj = Clock::now();
    if (pending_loop.active && fi >= pending_loop.frame + 2) join_lane();  // its 'current' pyramid is about to be reused
    t_loopjoin += since(tlj);
    const auto tm0 = Clock::now();
    FramePacket pkt = feeder.next();
    clk.feed_wait += since(tm0);
    const StepOut& step = pkt.step;
    // pyramids of finished frames go back to the tracker lane; a pending verification still reads its keyframe's
    auto release_frames = [&](int done) { feeder.release_upto(pending_loop.active ? std::min(done, pending_loop.frame - 1) : done); };
    if (step.prev_pts.empty()) {  // first keyframe (T:1715-1733)
      Keyframe kf(arena);
      kf.kf_id = (int)kfs.size();
      kf.frame_idx = fi;
      kf.img_name = meta[(size_t)fi].name;
      kf.pose = cur;
      kf_desc.push_back(pkt.desc);
      keep_corners(fi, pkt.corners);
      for (const Track& tr : pkt.tracks) {
        kf.obs.emplace(tr.id, tr.p);
        track_hist.try_emplace(tr.id, arena).first->second.obs.push_back({kf.kf_id, tr.p});
      }
      kfs.push_back(std::move(kf));
      last_kf_frame = fi;
      emit(fi);
      release_frames(fi);
      t_first += since(tm0);
      continue;
    }
    const std::vector<V2>& p_i = step.prev_pts;
    const std::vector<V2>& p_j = step.cur_pts;
    const auto tm1 = Clock::now();
    std::optional<RelPose> rel;
    if (pkt.rel.valid()) rel = ransac_finish(*pkt.rel.get());                     // T:1739, started ahead on lane A
    else rel = find_E_ransac_gpu(ctx, K, p_i, p_j, 2500, 1e-3, 60, &clk, rcomm);
    clk.m_ransac += since(tm1);
    const auto tp0 = Clock::now();
    int inliers = 0;
    double parallax = 0.0;
    if (rel) {
      inliers = (int)rel->inliers.size();
      std::vector<double> ds;
      ds.reserve(rel->inliers.size());
      for (int idx : rel->inliers) ds.push_back(std::hypot(p_j[(size_t)idx].x - p_i[(size_t)idx].x, p_j[(size_t)idx].y - p_i[(size_t)idx].y));
      if (!ds.empty()) {
        std::nth_element(ds.begin(), ds.begin() + (long)(ds.size() / 2), ds.end());
        parallax = ds[ds.size() / 2];
      }
      cur = compose_right_inv(cur, rel->R_ji, rel->t_ji);  // T:1762
    }
    bool make_kf = true;  // T:1700-1704,1765
    if (!kfs.empty() && rel.has_value()) {
      if (fi - last_kf_frame < cfg.kf_min_gap) make_kf = false;
      else if (inliers < cfg.kf_min_inliers) make_kf = true;
      else make_kf = parallax >= cfg.kf_parallax_px;
    }
    t_par += since(tp0);
    const auto tm2 = Clock::now();
    if (make_kf) {
      // Edges stay in keyframe order (lane C first).  BA(k-1) refined the poses the triangulation below reads, but
      // nothing before it does: unless a loop closure has to be finished first, lane B is joined as late as that.
      // The verdict of the last keyframe's loop verification (lane C: KLT + RANSAC, ~260 us) is what the geometry thread waited
      // for longest per keyframe; everything that neither the verdict nor finish_loop() can change runs BEFORE the join now:
      // this keyframe's observation table and track histories, the correspondences of its sequential edge, the walk that
      // collects the triangulation jobs.  What finish_loop() reads or orders -- the map's observation lists (its BA gathers
      // from them) and the list of pending edges (the loop edge goes in front of this keyframe's sequential edge) -- is touched
      // only after the join.  SFMX_JOIN_C_EARLY=1: the join first, as before (A/B and tests; identical output).
      const bool late_join = std::getenv("SFMX_JOIN_C_EARLY") == nullptr;
      // Loop-closure detection and verification of THIS keyframe (T:1822-1866).  Neither reads anything the rest of the block
      // produces -- the candidate comes from the global descriptors, the correspondences from the two images and the old
      // keyframe's corner sequence -- so the verification is handed to lane C before everything else (with its own lane for the
      // edges: verify_early) and has the whole keyframe block to finish in; its verdict is consumed at the next keyframe as before.
      // new_id / n_after: this keyframe's id and the keyframe count once it has been pushed.
      auto detect_loop = [&](int new_id, int n_after, PendingLoop& dst) {
        const std::vector<float>& desc_new = pkt.desc;
        int best_id = -1;
        float best_score = 0.0f;
        for (int kk = 0; kk < n_after - 6; ++kk) {
          const float s = dot_desc(kf_desc[(size_t)kk], desc_new);
          if (s > best_score) { best_score = s; best_id = kk; }
        }
        if (!(best_id >= 0 && best_score > 0.94f)) return;
        const Keyframe& old_kf = kfs[(size_t)best_id];
        LKConfig lc = cfg.klt;
        lc.max_tracks = loop_corners;
        lc.min_tracks = 600;
        // corners of the old keyframe image (T:1841): a prefix of the sequence found when that frame was current (same
        // image, quality, min_dist); detected here only if the tracker never replenished on that frame
        std::vector<V2> pts0;
        {
          const auto ts0 = Clock::now();
          clk.shi_calls++;
          auto itc = kf_corners.find(old_kf.frame_idx);
          if (itc != kf_corners.end() && itc->second->serves(lc.max_tracks, lc.quality, lc.min_distance)) {
            clk.shi_memo_hits++;
            pts0 = itc->second->prefix(lc.max_tracks);
          } else {
            if (!old_pyr) check(ctx, sfmx_pyramid_create(ctx, w, h, feeder.levels_total(), &old_pyr), "pyramid_create");
            src.load(ctx, old_kf.frame_idx, old_pyr);
            pts0 = geo_det.detect(old_pyr, lc.max_tracks, lc.quality, lc.min_distance);
            kf_corners[old_kf.frame_idx] = std::make_shared<const CornerMemo>(
                CornerMemo{lc.quality, lc.min_distance, lc.max_tracks, (int)pts0.size() < lc.max_tracks, pts0});
          }
          clk.shi += since(ts0);
        }
        auto cell = std::make_shared<std::shared_ptr<RansacAhead>>();
        dst = PendingLoop{true, fi, old_kf.kf_id, new_id, cell, std::nullopt, 0};
        const int old_frame = old_kf.frame_idx;
        const sfmx_pyramid* cur_pyr = pkt.pyr;  // not released to the tracker lane while this verification is pending
        auto verify = [&, cell, lc, old_frame, cur_pyr, pts0 = std::move(pts0)]() {
          sfmx_pyramid* opc = old_pyr_c;
          if (lane_c) opc = lane_c->pooled()->pyramid(w, h, feeder.levels_total());  // lives with the pooled context
          else if (!opc) { check(cctx, sfmx_pyramid_create(cctx, w, h, feeder.levels_total(), &old_pyr_c), "pyramid_create"); opc = old_pyr_c; }
          src.load(cctx, old_frame, opc);
          std::vector<V2> fwd;
          std::vector<std::uint8_t> keep;
          klt_pairs(cctx, lc, opc, cur_pyr, pts0, fwd, keep, cclk);
          std::vector<V2> li, lj;
          for (size_t i = 0; i < pts0.size(); i++) {
            if (!keep[i]) continue;
            li.push_back(pts0[i]);
            lj.push_back(fwd[i]);
          }
          if (li.size() >= 120) *cell = ransac_ahead(cctx, K, li, lj, 4000, 2e-3, 80, cclk);
        };
        if (lane_c) dst.ticket = lane_c->submit_ticket(std::move(verify));
        else { verify(); join_lane(); }
      };
      if (verify_early) detect_loop((int)kfs.size(), (int)kfs.size() + 1, queued_loop);
      bool looped = false;
      if (!late_join) {
        looped = join_c();
        if (looped) { join_b(); finish_loop(); }
      }
      Keyframe kf(arena);
      kf.kf_id = (int)kfs.size();
      kf.frame_idx = fi;
      kf.img_name = meta[(size_t)fi].name;
      kf.pose = cur;
      const std::vector<float>& new_desc = pkt.desc;
      keep_corners(fi, pkt.corners);
      const auto tb0 = Clock::now();
      std::vector<std::pair<int, V2>> late_obs;  // observations of already mapped tracks: into the map after the join
      if (late_join) late_obs.reserve(pkt.tracks.size());
      for (const Track& tr : pkt.tracks) {
        kf.obs.emplace(tr.id, tr.p);
        TrackHist& th = track_hist.try_emplace(tr.id, arena).first->second;
        th.obs.push_back({kf.kf_id, tr.p});
        if (th.mapped) {
          if (late_join) late_obs.emplace_back(tr.id, tr.p);
          else map.add_obs(tr.id, kf.kf_id, tr.p);
        }
      }
      std::vector<V2> ei, ej;
      int edge_prev_id = -1;
      if (!kfs.empty()) {  // sequential pose-graph edge (T:1782-1798): its correspondences
        const Keyframe& prev_kf = kfs.back();
        edge_prev_id = prev_kf.kf_id;
        ei.reserve(1200);
        ej.reserve(1200);
        for (const auto& kv : kf.obs) {
          auto itp = prev_kf.obs.find(kv.first);
          if (itp == prev_kf.obs.end()) continue;
          ei.push_back(itp->second);
          ej.push_back(kv.second);
        }
      }
      clk.bookkeeping += since(tb0);
      // triangulation jobs (T:1801-1813).  The DLT solves (libm Jacobi) are independent: gather the jobs in the reference's
      // iteration order, solve them on the host pool, then insert into the map sequentially in that same order.
      struct TriJob { int tid; TrackHist* th; const ObsList* hist; const Pose* p0; const Pose* pl; V3 X; bool ok; };
      std::vector<TriJob> jobs;
      const auto th0 = Clock::now();
      if (kfs.size() >= 1) {
        for (auto& kv : track_hist) {
          const int tid = kv.first;
          auto& hist = kv.second.obs;
          if (kv.second.mapped || hist.size() < 2) continue;
          const int id0 = hist.front().first, idl = hist.back().first;
          if (id0 == idl) continue;
          // Quirk Q12: the reference indexes kfs[idl] at T:1809 before the keyframe under
          // construction is pushed (T:1815), i.e. one past the end of the vector whenever idl is the
          // new keyframe — undefined behaviour whose result is heap garbage.  The defined reading is
          // used here: the new keyframe's own pose.
          const Pose* pl = (idl < (int)kfs.size()) ? &kfs[(size_t)idl].pose : &kf.pose;
          jobs.push_back(TriJob{tid, &kv.second, &hist, &kfs[(size_t)id0].pose, pl, V3{}, true});
        }
      }
      clk.tri_iter += since(th0);
      clk.host += since(th0);
      if (late_join) {
        looped = join_c();
        if (looped) { join_b(); finish_loop(); }
        for (const auto& ob : late_obs) map.add_obs(ob.first, kf.kf_id, ob.second);
      }
      if (edge_prev_id >= 0 && ei.size() >= 80) {
        pending_edges.push_back(PendingEdge{edge_prev_id, kf.kf_id, nullptr});
        PendingEdge* slot = &pending_edges.back();  // std::deque: stays valid while later edges are appended
        auto task = [slot, ectx, eclk, K, ransac_ahead, ei = std::move(ei), ej = std::move(ej)]() {
          slot->ra = ransac_ahead(ectx, K, ei, ej, 2500, 1e-3, 60, eclk);
        };
        if (edge_lane) edge_lane->submit(std::move(task));
        else task();
      }
      if (kfs.size() >= 1) {
        // the walk above only reads the track histories; the solves below read the keyframe poses BA(k-1) refines
        if (!looped) join_b();
        const auto ts0 = Clock::now();
        ThreadPool::instance().parallel_for((int)jobs.size(), [&](int i) {
          TriJob& j = jobs[(size_t)i];
          j.ok = triangulate_dlt(K, *j.p0, *j.pl, j.hist->front().second, j.hist->back().second, j.X);
        });
        clk.tri_solve += since(ts0);
        const auto ti0 = Clock::now();
        for (const TriJob& j : jobs) {
          if (!j.ok) throw std::runtime_error("Singular K");
          map.add(j.tid, j.X);
          j.th->mapped = true;
          for (const auto& ob : *j.hist) map.add_obs(j.tid, ob.first, ob.second);
        }
        clk.tri_insert += since(ti0);
        clk.host += since(ts0);
      } else if (!looped) {
        join_b();
      }
      kfs.push_back(std::move(kf));
      kf_desc.push_back(new_desc);
      last_kf_frame = fi;
      // T:1820: local BA.  The point/observation gather reads kfs and map now; the device iterations refine a
      // private copy of the window poses on lane B and are written back at the next join.
      const auto tg0 = Clock::now();
      pending_ba = GpuBundleAdjuster::gather(K, kfs, map, cfg.ba);
      clk.ba_gather += since(tg0);
      if (lane) lane->submit([&ba, &pending_ba]() { ba.solve(pending_ba); });
      else ba.solve(pending_ba);
      if (!lane) join_lane();

      // loop closure (T:1822-1866): detection + verification were started at the top of this block (detect_loop), or start here
      if (verify_early) {
        pending_loop = queued_loop;
        queued_loop = PendingLoop{};
      } else {
        detect_loop(kfs.back().kf_id, (int)kfs.size(), pending_loop);
      }
    }
    clk.m_kf += since(tm2);
    const auto te0 = Clock::now();
    emit(fi);
    t_emit += since(te0);
    const auto tr0 = Clock::now();
    release_frames(fi);
    t_rel += since(tr0);
  }
  if (trace_phases) std::fprintf(stderr, "loop parts: parallax %.3f emit %.3f release %.3f first-frame %.3f loop-join %.3f feed %.3f ransac %.3f kf %.3f (join B %.3f, join C %.3f) ms\n", t_par * 1e3, t_emit * 1e3, t_rel * 1e3, t_first * 1e3, t_loopjoin * 1e3, clk.feed_wait * 1e3, clk.m_ransac * 1e3, clk.m_kf * 1e3, t_join_b * 1e3, (clk.join_wait - t_join_b) * 1e3);
  phase("frame loop done");
  join_lane();
  flush_edges();
  phase("lanes joined");
  feeder.finish();
  phase("feeder finished");
  if (lane_a2) {
    lane_a2->wait();
    lane_a_clk.add(lane_a2_clk);
    if (sfmx_get_timing(ctx)) clk.grab_profile(lane_a2->ctx());
  }
  if (lane_a) {
    lane_a->wait();
    clk.lane_a_busy = std::max(lane_a->busy_seconds(), lane_a2 ? lane_a2->busy_seconds() : 0.0);
    clk.ransac += lane_a_clk.ransac; clk.ransac_kernel_us += lane_a_clk.ransac_kernel_us; clk.ransac_calls += lane_a_clk.ransac_calls;
    clk.ransac_points += lane_a_clk.ransac_points; clk.ransac_verified += lane_a_clk.ransac_verified;
    clk.ransac_cert_misses += lane_a_clk.ransac_cert_misses;
    clk.r_pre += lane_a_clk.r_pre; clk.r_gpu += lane_a_clk.r_gpu; clk.r_verify += lane_a_clk.r_verify; clk.r_decomp += lane_a_clk.r_decomp;
    if (sfmx_get_timing(ctx)) clk.grab_profile(lane_a->ctx());
  }
  if (sfmx_get_timing(ctx)) {  // per-kernel profiles of every context that worked on this run
    clk.grab_profile(ctx);
    if (feeder.threaded()) clk.grab_profile(feeder.ctx());
    if (lane) clk.grab_profile(lane->ctx());
    if (lane_c) clk.grab_profile(lane_c->ctx());
    if (lane_e) clk.grab_profile(lane_e->ctx());
    if (prefetch) prefetch->grab_profile(clk);
  }
  if (feeder.threaded()) clk.add(feeder.lane_clock());
  if (lane) clk.lane_b_busy = lane->busy_seconds();
  if (lane_c) clk.lane_c_busy = lane_c->busy_seconds();
  if (lane_c) {
    lane_clk.ransac += lane_c_clk.ransac; lane_clk.ransac_kernel_us += lane_c_clk.ransac_kernel_us; lane_clk.ransac_calls += lane_c_clk.ransac_calls;
    lane_clk.ransac_points += lane_c_clk.ransac_points; lane_clk.ransac_verified += lane_c_clk.ransac_verified;
    lane_clk.ransac_cert_misses += lane_c_clk.ransac_cert_misses;
    lane_clk.r_pre += lane_c_clk.r_pre; lane_clk.r_gpu += lane_c_clk.r_gpu; lane_clk.r_verify += lane_c_clk.r_verify; lane_clk.r_decomp += lane_c_clk.r_decomp;
    clk.klt += lane_c_clk.klt; clk.klt_kernel_us += lane_c_clk.klt_kernel_us; clk.lk_steps += lane_c_clk.lk_steps;
    clk.tracks_in += lane_c_clk.tracks_in; clk.klt_calls += lane_c_clk.klt_calls;
  }
  if (lane_e) {
    clk.lane_e_busy = lane_e->busy_seconds();
    lane_clk.ransac += lane_e_clk.ransac; lane_clk.ransac_kernel_us += lane_e_clk.ransac_kernel_us; lane_clk.ransac_calls += lane_e_clk.ransac_calls;
    lane_clk.ransac_points += lane_e_clk.ransac_points; lane_clk.ransac_verified += lane_e_clk.ransac_verified;
    lane_clk.ransac_cert_misses += lane_e_clk.ransac_cert_misses;
    lane_clk.r_pre += lane_e_clk.r_pre; lane_clk.r_gpu += lane_e_clk.r_gpu; lane_clk.r_verify += lane_e_clk.r_verify; lane_clk.r_decomp += lane_e_clk.r_decomp;
  }
  if (lane) {  // lane B's (and, merged above, lane C's and E's) counters
    clk.ba += lane_clk.ba; clk.ba_kernel_us += lane_clk.ba_kernel_us; clk.ba_calls += lane_clk.ba_calls; clk.ba_iters += lane_clk.ba_iters;
    clk.ransac += lane_clk.ransac; clk.ransac_kernel_us += lane_clk.ransac_kernel_us; clk.ransac_calls += lane_clk.ransac_calls;
    clk.ransac_points += lane_clk.ransac_points; clk.ransac_verified += lane_clk.ransac_verified;
    clk.ransac_cert_misses += lane_clk.ransac_cert_misses;
    clk.r_pre += lane_clk.r_pre; clk.r_gpu += lane_clk.r_gpu; clk.r_verify += lane_clk.r_verify; clk.r_decomp += lane_clk.r_decomp;
  }
  out.log = so.str();
  if (prefetch) {
    clk.shi_fallbacks += prefetch->replays();
    clk.shi_kernel_us += prefetch->kernel_us();
    prefetch->busy(clk.pf_busy, clk.pf_gpu, clk.pf_replay);  // the worker's tie-order replays count too
    prefetch.reset();
  }
  phase("prefetch torn down");
  clk.total = since(t_all);
}

void write_outputs(const std::string& out_dir, const PipelineConfig& cfg, const std::vector<FrameMeta>& meta, PipelineResult& out) {
  namespace fs = std::filesystem;
  const fs::path dir(out_dir);
  fs::create_directories(dir);
  {
    std::ofstream f(dir / "keyframes_camera_centers.csv");  // T:1463-1475
    f << "kf_id,frame_idx,image,x,y,z,lat,lon\n";
    for (const Keyframe& kf : out.kfs) {
      const FrameMeta& m = meta[(size_t)kf.frame_idx];
      f << kf.kf_id << "," << kf.frame_idx << "," << kf.img_name << "," << kf.pose.t.x << "," << kf.pose.t.y << "," << kf.pose.t.z << ","
        << (m.has_ang ? m.lat : 0.0) << "," << (m.has_ang ? m.lon : 0.0) << "\n";
    }
  }
  {
    std::ofstream f(dir / "posegraph_edges.csv");  // T:1199-1209
    f << "i,j,rvec_x,rvec_y,rvec_z,t_x,t_y,t_z,inliers,is_loop\n";
    for (const PGEdge& e : out.edges) {
      const V3 rv = so3_log(e.R_ji);
      f << e.i << "," << e.j << "," << rv.x << "," << rv.y << "," << rv.z << "," << e.t_ji.x << "," << e.t_ji.y << "," << e.t_ji.z << ","
        << e.inliers << "," << (e.is_loop ? 1 : 0) << "\n";
    }
  }
  if (cfg.export_pointcloud) {  // T:1215-1224,1878-1883
    const fs::path p = dir / "templeRing_sparse_points.ply";
    std::ofstream f(p);
    if (!f) throw std::runtime_error("Failed to write: " + p.string());
    f << "ply\nformat ascii 1.0\n";
    f << "element vertex " << out.map.pts.size() << "\n";
    f << "property float x\nproperty float y\nproperty float z\nend_header\n";
    for (const auto& kv : out.map.pts) f << kv.second.Xw.x << " " << kv.second.Xw.y << " " << kv.second.Xw.z << "\n";
  }
  std::ostringstream so;
  so << "\n=== Summary ===\n";
  so << "Keyframes: " << out.kfs.size() << "\n";
  so << "Map points: " << out.map.pts.size() << "\n";
  so << "Outputs: " << dir << "\n";
  out.log += so.str();
}

}  // namespace sfmx_host

// ============================================================================================ C ABI
// Used by bench.py / tests through ctypes (libsfmx_host.so).  Plain pointers and sizes only.
extern "C" {

struct sfmx_pipeline_cfg {
  int frames, export_pointcloud;
  int max_tracks, min_tracks;
  double quality;
  int min_distance, pyr_levels, win_radius, klt_iters;
  double fb_thresh;
  int kf_min_gap, kf_min_inliers;
  double kf_parallax_px;
  int ba_window, ba_iters, ba_max_points;
  double ba_huber, ba_lambda;
  sfmx_comm *comm_ba, *comm_ransac;  // multi-GPU mode (PipelineConfig); null = unsharded
};
struct sfmx_pipeline_stats {
  int n_keyframes, n_points, n_edges, n_frames;
  double sec_total, sec_klt, sec_shi, sec_ransac, sec_ba, sec_upload, sec_host, sec_shi_gpu, sec_shi_replay, sec_desc, sec_bookkeeping, sec_r_pre, sec_r_gpu, sec_r_verify, sec_r_decomp, sec_tri_iter, sec_tri_solve, sec_tri_insert;
  double us_klt_kernel, us_ransac_kernel, us_ba_kernel, us_shi_kernel;
  unsigned long long lk_steps, tracks_in, klt_calls, ransac_calls, ransac_points, ba_calls, ba_iters, ransac_verified, shi_fallbacks, shi_calls, shi_memo_hits, shi_prefetched;
  double sec_shi_wait, sec_setup, sec_wall;
  double sec_pf_busy, sec_pf_gpu, sec_pf_replay, sec_lane_b_busy, sec_lane_c_busy, sec_join_wait, sec_ba_gather;
  double sec_m_step, sec_m_ransac, sec_m_kf, sec_feed_wait;
  unsigned long long ransac_cert_misses;
  double us_kernel[16];
  unsigned long long calls_kernel[16];
  double sec_lane_a_busy, sec_lane_e_busy;
};

// images_host and/or images_dev: [n][h][w] u8 (images_dev = device pointer, frames already in HBM).
// out_dir may be NULL (no files).  centres_out (optional) [n_keyframes<=cap][3].
static int pipeline_run_body(sfmx_ctx* ctx, const std::uint8_t* images_host, const void* images_dev, int n_images, int w, int h,
                             const char* const* names, const double* K9, const double* lat, const double* lon, const std::uint8_t* has_ang,
                             const sfmx_pipeline_cfg* cfg, const char* out_dir, char* log, int log_cap, sfmx_pipeline_stats* stats,
                             double* centres_out, int centres_cap, sfmx_host::Clock::time_point t_wall) {
  using namespace sfmx_host;
  try {
    MemoryFrames src;
    src.host = images_host;
    src.dev = static_cast<const std::uint8_t*>(images_dev);
    src.n = n_images; src.w = w; src.h = h;
    std::vector<FrameMeta> meta((size_t)n_images);
    for (int i = 0; i < n_images; i++) {
      meta[(size_t)i].name = names ? names[i] : ("frame" + std::to_string(i));
      meta[(size_t)i].has_ang = has_ang ? has_ang[i] != 0 : false;
      meta[(size_t)i].lat = lat ? lat[i] : 0.0;
      meta[(size_t)i].lon = lon ? lon[i] : 0.0;
    }
    Mat3 K;
    std::memcpy(K.a, K9, 72);
    PipelineConfig pc;
    pc.frames = cfg->frames;
    pc.export_pointcloud = cfg->export_pointcloud != 0;
    pc.klt.max_tracks = cfg->max_tracks; pc.klt.min_tracks = cfg->min_tracks; pc.klt.quality = cfg->quality;
    pc.klt.min_distance = cfg->min_distance; pc.klt.pyr_levels = cfg->pyr_levels; pc.klt.win_radius = cfg->win_radius;
    pc.klt.iters = cfg->klt_iters; pc.klt.fb_thresh = cfg->fb_thresh;
    pc.kf_min_gap = cfg->kf_min_gap; pc.kf_min_inliers = cfg->kf_min_inliers; pc.kf_parallax_px = cfg->kf_parallax_px;
    pc.ba.window = cfg->ba_window; pc.ba.iters = cfg->ba_iters; pc.ba.max_points = cfg->ba_max_points;
    pc.ba.huber_delta = cfg->ba_huber; pc.ba.lambda = cfg->ba_lambda;
    pc.comm_ba = cfg->comm_ba; pc.comm_ransac = cfg->comm_ransac;
    PipelineResult res;
    if (std::getenv("SFMX_TRACE_PHASES")) std::fprintf(stderr, "phase %-22s %8.3f ms (since entry)\n", "inputs wrapped", since(t_wall) * 1e3);
    run_pipeline(ctx, src, meta, K, pc, res);  // returns after its lanes / prefetch contexts are torn down
    const double wall = since(t_wall);
    if (std::getenv("SFMX_TRACE_PHASES")) std::fprintf(stderr, "phase %-22s %8.3f ms (since entry)\n", "run_pipeline returned", wall * 1e3);
    if (out_dir) write_outputs(out_dir, pc, meta, res);
    if (log && log_cap > 0) std::snprintf(log, (size_t)log_cap, "%s", res.log.c_str());
    if (stats) {
      const StageClock& c = res.clock;
      *stats = sfmx_pipeline_stats{(int)res.kfs.size(), (int)res.map.pts.size(), (int)res.edges.size(), std::min(pc.frames, n_images),
                                   c.total, c.klt, c.shi, c.ransac, c.ba, c.upload, c.host, c.shi_gpu, c.shi_replay, c.desc, c.bookkeeping, c.r_pre, c.r_gpu, c.r_verify, c.r_decomp, c.tri_iter, c.tri_solve, c.tri_insert,
                                   c.klt_kernel_us, c.ransac_kernel_us, c.ba_kernel_us, c.shi_kernel_us,
                                   c.lk_steps, c.tracks_in, c.klt_calls, c.ransac_calls, c.ransac_points, c.ba_calls, c.ba_iters, c.ransac_verified, c.shi_fallbacks, c.shi_calls, c.shi_memo_hits, c.shi_prefetched, c.shi_wait, c.setup, wall,
                                   c.pf_busy, c.pf_gpu, c.pf_replay, c.lane_b_busy, c.lane_c_busy, c.join_wait, c.ba_gather,
                                   c.m_step, c.m_ransac, c.m_kf, c.feed_wait, c.ransac_cert_misses, {}, {}};
      for (int i = 0; i < 16; i++) { stats->us_kernel[i] = c.kernel_us[i]; stats->calls_kernel[i] = c.kernel_calls[i]; }
      stats->sec_lane_a_busy = c.lane_a_busy;
      stats->sec_lane_e_busy = c.lane_e_busy;
    }
    if (centres_out)
      for (int k = 0; k < (int)res.kfs.size() && k < centres_cap; k++) {
        centres_out[3 * k] = res.kfs[(size_t)k].pose.t.x;
        centres_out[3 * k + 1] = res.kfs[(size_t)k].pose.t.y;
        centres_out[3 * k + 2] = res.kfs[(size_t)k].pose.t.z;
      }
    return SFMX_OK;
  } catch (const sfmx_host::SfmxFailure& e) {
    if (log && log_cap > 0) std::snprintf(log, (size_t)log_cap, "ERROR: %s\n", e.what());
    return e.status;
  } catch (const std::exception& e) {
    if (log && log_cap > 0) std::snprintf(log, (size_t)log_cap, "ERROR: %s\n", e.what());
    return SFMX_ERR_INVALID;
  }
}

int sfmx_pipeline_run(sfmx_ctx* ctx, const std::uint8_t* images_host, const void* images_dev, int n_images, int w, int h,
                      const char* const* names, const double* K9, const double* lat, const double* lon, const std::uint8_t* has_ang,
                      const sfmx_pipeline_cfg* cfg, const char* out_dir, char* log, int log_cap, sfmx_pipeline_stats* stats,
                      double* centres_out, int centres_cap) {
  using namespace sfmx_host;
  if (!ctx || !cfg || !K9 || (!images_host && !images_dev) || n_images <= 0) return SFMX_ERR_INVALID;
  const auto t_wall = Clock::now();
  const int rc = pipeline_run_body(ctx, images_host, images_dev, n_images, w, h, names, K9, lat, lon, has_ang, cfg, out_dir, log, log_cap, stats,
                                   centres_out, centres_cap, t_wall);
  // the body's locals (map, keyframes, track histories) are gone here
  if (std::getenv("SFMX_TRACE_PHASES")) std::fprintf(stderr, "phase %-22s %8.3f ms (since entry)\n", "results released", since(t_wall) * 1e3);
  return rc;
}

// frees the helper contexts kept for reuse by later sfmx_pipeline_run calls
void sfmx_host_release_contexts() { sfmx_host::ContextPool::instance().clear(); }

// find_E_ransac seam (T:646-761) on its own, for the parity tests: 1 = pose found, 0 = none (T:648, T:678), < 0 = -status
int sfmx_host_find_E_ransac(sfmx_ctx* ctx, const double* K9, const double* pi, const double* pj, int n, int iters, double thr, int min_inliers,
                            double* R9, double* t3, int* inliers, int* n_inl, int* best_iter) {
  using namespace sfmx_host;
  try {
    Mat3 K;
    std::memcpy(K.a, K9, 72);
    std::vector<V2> a((size_t)std::max(n, 0)), b((size_t)std::max(n, 0));
    for (int i = 0; i < n; i++) { a[(size_t)i] = {pi[2 * i], pi[2 * i + 1]}; b[(size_t)i] = {pj[2 * i], pj[2 * i + 1]}; }
    const auto r = find_E_ransac_gpu(ctx, K, a, b, iters, thr, min_inliers, nullptr);
    *n_inl = 0;
    if (!r) return 0;
    std::memcpy(R9, r->R_ji.a, 72);
    t3[0] = r->t_ji.x; t3[1] = r->t_ji.y; t3[2] = r->t_ji.z;
    *n_inl = (int)r->inliers.size();
    for (size_t i = 0; i < r->inliers.size(); i++) inliers[i] = r->inliers[i];
    if (best_iter) *best_iter = r->best_iter;
    return 1;
  } catch (const SfmxFailure& e) {
    return -e.status;
  } catch (const std::exception&) {
    return -SFMX_ERR_INVALID;
  }
}

// The same seam as `world` ranks would run it (TEST HOOK, one GPU): every virtual rank's half (ransac_local on its iteration
// range), the merge the two all-reduce(max) compute -- largest packed (count, ~iteration) key, that rank's E as raw bits --
// and the result as rank `as_rank` forms it: its own mask if it holds the winner, otherwise the mask recomputed from the E bits.
int sfmx_host_find_E_ransac_world(sfmx_ctx* ctx, const double* K9, const double* pi, const double* pj, int n, int iters, double thr,
                                  int min_inliers, int world, int as_rank, double* R9, double* t3, int* inliers, int* n_inl, int* best_iter) {
  using namespace sfmx_host;
  try {
    if (world < 1 || as_rank < 0 || as_rank >= world) return -SFMX_ERR_INVALID;
    Mat3 K;
    std::memcpy(K.a, K9, 72);
    std::vector<V2> a((size_t)std::max(n, 0)), b((size_t)std::max(n, 0));
    for (int i = 0; i < n; i++) { a[(size_t)i] = {pi[2 * i], pi[2 * i + 1]}; b[(size_t)i] = {pj[2 * i], pj[2 * i + 1]}; }
    std::vector<RansacLocal> loc;
    std::uint64_t best_key = 0;
    int best_rank = -1;
    for (int r = 0; r < world; r++) {
      loc.push_back(ransac_local(ctx, K, a, b, iters, thr, min_inliers, nullptr, r, world));
      const RansacLocal& l = loc.back();
      const std::uint64_t key = l.win_iter >= 0 ? (((std::uint64_t)(std::uint32_t)l.win_count << 32) | (std::uint64_t)(0x7fffffff - l.win_iter)) : 0;
      if (key > best_key) { best_key = key; best_rank = r; }
    }
    RansacLocal mine = std::move(loc[(size_t)as_rank]);
    if (best_rank < 0) { mine.win_iter = -1; mine.win_count = -1; }
    else if (best_rank != as_rank) {
      const RansacLocal& w = loc[(size_t)best_rank];
      mine.win_iter = w.win_iter;
      mine.win_count = w.win_count;
      mine.winE = w.winE;  // travels as raw bits
      mine.win_mask.assign((size_t)mine.n, 0);
      std::int32_t cnt = 0;
      if (mine.n > 0 && sfmx_sampson_mask(ctx, mine.xi.data(), mine.xj.data(), mine.n, mine.winE.a, mine.thr, mine.win_mask.data(), &cnt) != SFMX_OK)
        return -SFMX_ERR_HIP;
    }
    const auto r = ransac_merge(ctx, nullptr, std::move(mine), nullptr);
    *n_inl = 0;
    if (!r) return 0;
    std::memcpy(R9, r->R_ji.a, 72);
    t3[0] = r->t_ji.x; t3[1] = r->t_ji.y; t3[2] = r->t_ji.z;
    *n_inl = (int)r->inliers.size();
    for (size_t i = 0; i < r->inliers.size(); i++) inliers[i] = r->inliers[i];
    if (best_iter) *best_iter = r->best_iter;
    return 1;
  } catch (const SfmxFailure& e) {
    return -e.status;
  } catch (const std::exception&) {
    return -SFMX_ERR_INVALID;
  }
}

// KLTTracker seam (T:307-400) on its own, one image per step, for the parity tests against the reference's tracker state
struct sfmx_host_tracker {
  struct OneImage : sfmx_host::FrameSource {
    const std::uint8_t* pix = nullptr;
    int w = 0, h = 0;
    int count() const override { return 1 << 30; }
    int width() const override { return w; }
    int height() const override { return h; }
    void load(sfmx_ctx* ctx, int, sfmx_pyramid* pyr) override {
      if (sfmx_pyramid_upload(ctx, pyr, pix) != SFMX_OK) throw std::runtime_error("pyramid_upload");
    }
  } src;
  std::unique_ptr<sfmx_host::GpuTracker> trk;
  int frame = 0;
};
void* sfmx_host_tracker_create(sfmx_ctx* ctx, int w, int h, int max_tracks, int min_tracks, double quality, int min_distance, int levels,
                               int radius, int iters, double fb) {
  try {
    auto t = std::make_unique<sfmx_host_tracker>();
    t->src.w = w; t->src.h = h;
    sfmx_host::LKConfig c;
    c.max_tracks = max_tracks; c.min_tracks = min_tracks; c.quality = quality; c.min_distance = min_distance;
    c.pyr_levels = levels; c.win_radius = radius; c.iters = iters; c.fb_thresh = fb;
    t->trk = std::make_unique<sfmx_host::GpuTracker>(ctx, c, w, h, 0, nullptr);
    return t.release();
  } catch (...) {
    return nullptr;
  }
}
void sfmx_host_tracker_destroy(void* h) { delete static_cast<sfmx_host_tracker*>(h); }
// StepOut of KLTTracker::step (T:340-391): returns the number of survivors (prev/cur [n][2], ids [n]), < 0 on error
int sfmx_host_tracker_step(void* h, const std::uint8_t* pix, double* prev_xy, double* cur_xy, int* ids, int cap) {
  auto* t = static_cast<sfmx_host_tracker*>(h);
  try {
    t->src.pix = pix;
    const sfmx_host::StepOut o = t->trk->step(t->src, t->frame++);
    const int n = (int)o.ids.size();
    if (n > cap) return -SFMX_ERR_INVALID;
    for (int i = 0; i < n; i++) {
      prev_xy[2 * i] = o.prev_pts[(size_t)i].x; prev_xy[2 * i + 1] = o.prev_pts[(size_t)i].y;
      cur_xy[2 * i] = o.cur_pts[(size_t)i].x; cur_xy[2 * i + 1] = o.cur_pts[(size_t)i].y;
      ids[i] = o.ids[(size_t)i];
    }
    return n;
  } catch (const sfmx_host::SfmxFailure& e) {
    return -e.status;
  } catch (const std::exception&) {
    return -SFMX_ERR_INVALID;
  }
}
int sfmx_host_tracker_tracks(void* h, double* xy, int* ids, int cap) {  // KLTTracker::tracks() (T:393)
  auto* t = static_cast<sfmx_host_tracker*>(h);
  const auto& tr = t->trk->tracks();
  const int n = (int)tr.size();
  if (n > cap) return -SFMX_ERR_INVALID;
  for (int i = 0; i < n; i++) { xy[2 * i] = tr[(size_t)i].p.x; xy[2 * i + 1] = tr[(size_t)i].p.y; ids[i] = tr[(size_t)i].id; }
  return n;
}

// posegraph_optimize_centers (T:1131-1197) on its own: R9s [n][9] camera->world rotations, centres [n][3] updated in
// place, edges (i, j, R_ji [9], t_ji [3], is_loop).  1 = solved, 0 = skipped (singular / empty), < 0 = -status
int sfmx_host_posegraph(sfmx_ctx* ctx, int n_kf, const double* R9s, double* centres3, int n_edges, const int* ei, const int* ej, const double* eR,
                        const double* et, const int* is_loop) {
  using namespace sfmx_host;
  try {
    Arena arena;
    std::vector<Keyframe> kfs;
    kfs.reserve((size_t)n_kf);
    for (int k = 0; k < n_kf; k++) {
      kfs.emplace_back(&arena);
      kfs.back().kf_id = k;
      std::memcpy(kfs.back().pose.R.a, R9s + 9 * k, 72);
      kfs.back().pose.t = {centres3[3 * k], centres3[3 * k + 1], centres3[3 * k + 2]};
    }
    std::vector<PGEdge> edges((size_t)n_edges);
    for (int e = 0; e < n_edges; e++) {
      edges[(size_t)e].i = ei[e];
      edges[(size_t)e].j = ej[e];
      std::memcpy(edges[(size_t)e].R_ji.a, eR + 9 * e, 72);
      edges[(size_t)e].t_ji = {et[3 * e], et[3 * e + 1], et[3 * e + 2]};
      edges[(size_t)e].inliers = 0;
      edges[(size_t)e].is_loop = is_loop[e] != 0;
    }
    const bool ok = posegraph_optimize_centers(ctx, kfs, edges);
    for (int k = 0; k < n_kf; k++) { centres3[3 * k] = kfs[(size_t)k].pose.t.x; centres3[3 * k + 1] = kfs[(size_t)k].pose.t.y; centres3[3 * k + 2] = kfs[(size_t)k].pose.t.z; }
    return ok ? 1 : 0;
  } catch (const SfmxFailure& e) {
    return -e.status;
  } catch (const std::exception&) {
    return -SFMX_ERR_INVALID;
  }
}

// the CLI's file readers on their own (csrc/host/cli_io.hpp), for the surface tests against the reference's readers
int sfmx_host_read_pgm(const char* path, int* w, int* h, unsigned long long* checksum, char* err, int cap) {
  try {
    const sfmx_cli::Gray im = sfmx_cli::read_pgm(path);
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
// kind 0 = int, 1 = double, 2 = string.  1 = found, 0 = absent or of another type, -1 = parse error (message in text_out)
int sfmx_host_config_lookup(const char* json, const char* section, const char* key, int kind, double* num_out, char* text_out, int cap) {
  try {
    const sfmx_cli::Json doc = sfmx_cli::Json::parse(json);
    const sfmx_cli::Json::Ref v = sfmx_cli::config_value(doc, section, key);
    if (kind == 0) { const auto r = sfmx_cli::as_int(v); if (!r) return 0; *num_out = *r; return 1; }
    if (kind == 1) { const auto r = sfmx_cli::as_number(v); if (!r) return 0; *num_out = *r; return 1; }
    const auto r = sfmx_cli::as_string(v);
    if (!r) return 0;
    std::snprintf(text_out, (size_t)cap, "%s", r->c_str());
    return 1;
  } catch (const std::exception& e) {
    std::snprintf(text_out, (size_t)cap, "%s", e.what());
    return -1;
  }
}

// host-side math self-checks used by the CPU test-suite (no device involved)
// Iteration order of an unordered_map<int,int> filled (and partly erased) with `keys`: default allocator vs the bump
// arena.  ops[i] != 0 erases keys[i] instead of inserting it.  Returns the number of live entries written to each output.
int sfmx_host_map_order(const int* keys, const unsigned char* ops, int n, int* order_default, int* order_arena) {
  std::unordered_map<int, int> a;
  sfmx_host::Arena arena;
  sfmx_host::ArenaMap<int, int> b(0, std::hash<int>(), std::equal_to<int>(), sfmx_host::ArenaAlloc<std::pair<const int, int>>(&arena));
  for (int i = 0; i < n; i++) {
    if (ops && ops[i]) { a.erase(keys[i]); b.erase(keys[i]); }
    else { a.emplace(keys[i], i); b.emplace(keys[i], i); }
  }
  int k = 0;
  for (const auto& kv : a) order_default[k++] = kv.first;
  int m = 0;
  for (const auto& kv : b) order_arena[m++] = kv.first;
  return k == m ? k : -1;
}
void sfmx_host_eight_point_E(const double* xn, const double* yn, const int* idx8, double* E9) {
  const sfmx_host::Mat3 E = sfmx_host::eight_point_E(xn, yn, idx8);
  std::memcpy(E9, E.a, 72);
}
void sfmx_host_uniform_draws(unsigned seed, int n, int count, int* out) {
  sfmx_host::Mt19937 g(seed);
  for (int i = 0; i < count; i++) out[i] = g.below((std::uint32_t)n);
}
void sfmx_host_decompose_E(const double* E9, const double* xi, const double* xj, const int* inl, int n_inl, double* R9, double* t3) {
  sfmx_host::Mat3 E, R;
  std::memcpy(E.a, E9, 72);
  sfmx_host::V3 t;
  std::vector<int> v(inl, inl + n_inl);
  sfmx_host::decompose_E(E, xi, xj, v, R, t);
  std::memcpy(R9, R.a, 72);
  t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
}
void sfmx_host_triangulate_dlt(const double* K9, const double* Ri, const double* ti, const double* Rj, const double* tj, const double* ui,
                               const double* uj, double* X3) {
  sfmx_host::Mat3 K;
  std::memcpy(K.a, K9, 72);
  sfmx_host::Pose a, b;
  std::memcpy(a.R.a, Ri, 72); a.t = {ti[0], ti[1], ti[2]};
  std::memcpy(b.R.a, Rj, 72); b.t = {tj[0], tj[1], tj[2]};
  sfmx_host::V3 X;
  sfmx_host::triangulate_dlt(K, a, b, {ui[0], ui[1]}, {uj[0], uj[1]}, X);
  X3[0] = X.x; X3[1] = X.y; X3[2] = X.z;
}
void sfmx_host_so3(const double* w3, double* R9, double* log3) {
  const sfmx_host::Mat3 R = sfmx_host::so3_exp({w3[0], w3[1], w3[2]});
  std::memcpy(R9, R.a, 72);
  const sfmx_host::V3 l = sfmx_host::so3_log(R);
  log3[0] = l.x; log3[1] = l.y; log3[2] = l.z;
}
double sfmx_host_hypot(double x, double y) { return sfmx::hypot_glibc(x, y); }

// introsort replica checks: mode 0 = real std::sort (T:286 predicate), 1 = full replay, 2 = selective
// replay with marks.  ids_out receives the element ids in array order afterwards.  returns 1 on success.
int sfmx_host_sort_order(const double* scores, const unsigned char* marks, int n, int mode, unsigned* ids_out) {
  std::vector<sfmx_host::SortKey> k((size_t)n);
  for (int i = 0; i < n; i++) k[(size_t)i] = sfmx_host::SortKey{scores[i], (std::uint32_t)i, marks ? (std::uint32_t)marks[i] : 0u};
  bool ok = true;
  if (mode == 0) std::sort(k.begin(), k.end(), [](const sfmx_host::SortKey& a, const sfmx_host::SortKey& b) { return a.s > b.s; });
  else if (mode == 1) ok = sfmx_host::introsort_replay_full(k);
  else ok = sfmx_host::introsort_replay_selective(k);
  for (int i = 0; i < n; i++) ids_out[i] = k[(size_t)i].id;
  return ok ? 1 : 0;
}

}  // extern "C"
