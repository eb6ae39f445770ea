// thread_pool.hpp — tiny persistent worker teams for the host-side libm-bound small solves (DLT
// triangulations, cheirality votes).  Work items are independent and their results are written to
// pre-sized slots, so the outcome is identical to the sequential loop for any thread count.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sfmx_host {

// One team of worker threads; parallel_for is used by one caller at a time.
class WorkerTeam {
 public:
  WorkerTeam() {
    int want = 0;
    if (const char* e = std::getenv("SFMX_HOST_THREADS")) want = std::atoi(e);
    // 8: the regions are short (tens of 2-5 us solves); every extra worker is one more wake-up per region, and a pipeline
    // already runs ten lane threads (measured on the bench: 8 -> 1 393 keyframes/s, 5 -> 1 366, 11 -> 1 377, 16 -> 1 335, 48 -> 660)
    if (want <= 0) want = std::min(8, std::max(1, (int)std::thread::hardware_concurrency() / 2));
    for (int i = 1; i < want; i++) workers_.emplace_back([this] { worker(); });
  }
  ~WorkerTeam() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  bool solo() const { return workers_.empty(); }
  // calls fn(i) for i in [0,n); the calling thread participates.  Only as many workers are woken as there are chunks
  // beyond the caller's own: most regions are a few dozen 2-5 us solves, and a wake-up costs more than a chunk.
  void parallel_for(int n, const std::function<void(int)>& fn, int grain) {
    const int chunks = (n + grain - 1) / grain;
    const int need = std::min((int)workers_.size(), std::max(0, chunks - 1));
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_ = n;
      grain_ = grain;
      next_.store(0);
      tickets_ = need;
      pending_ = need;
    }
    if (need == (int)workers_.size()) cv_.notify_all();
    else for (int i = 0; i < need; i++) cv_.notify_one();
    run_chunks();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void run_chunks() {
    for (;;) {
      const int b = next_.fetch_add(grain_);
      if (b >= n_) break;
      const int e = std::min(n_, b + grain_);
      for (int i = b; i < e; i++) (*fn_)(i);
    }
  }
  void worker() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || tickets_ > 0; });  // a ticket = the right to join the current region
        if (stop_) return;
        --tickets_;
      }
      run_chunks();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, grain_ = 16, pending_ = 0, tickets_ = 0;
  std::atomic<int> next_{0};
  bool stop_ = false;
};

// Process-wide front end: every parallel region borrows a team for its duration, so that the lanes of one pipeline and
// several pipelines in one process (one sequence per host thread) never wait for each other; teams are created on
// demand and kept for reuse.
class ThreadPool {
 public:
  static ThreadPool& instance() {
    static ThreadPool* p = new ThreadPool;  // kept until process exit: no joins from static destructors
    return *p;
  }
  void parallel_for(int n, const std::function<void(int)>& fn, int grain = 16) {
    if (n <= 0) return;
    if (n <= grain) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    WorkerTeam* team = nullptr;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!idle_.empty()) {
        team = idle_.back();
        idle_.pop_back();
      }
    }
    if (!team) team = new WorkerTeam;
    if (team->solo()) {
      for (int i = 0; i < n; i++) fn(i);
    } else {
      team->parallel_for(n, fn, grain);
    }
    std::lock_guard<std::mutex> lk(mu_);
    idle_.push_back(team);
  }

 private:
  std::mutex mu_;
  std::vector<WorkerTeam*> idle_;
};

}  // namespace sfmx_host
