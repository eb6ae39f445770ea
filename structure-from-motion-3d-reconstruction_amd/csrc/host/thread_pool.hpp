// thread_pool.hpp — a tiny persistent pool for the host-side libm-bound small solves (DLT
// triangulations, cheirality votes).  Work items are independent and their results are written to
// pre-sized slots, so the outcome is identical to the sequential loop for any thread count.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace sfmx_host {

class ThreadPool {
 public:
  static ThreadPool& instance() {
    static ThreadPool p;
    return p;
  }
  int size() const { return (int)workers_.size() + 1; }
  // calls fn(i) for i in [0,n); the calling thread participates.
  void parallel_for(int n, const std::function<void(int)>& fn, int grain = 16) {
    if (n <= 0) return;
    // one parallel region at a time: a second caller (another pipeline lane) simply runs its items serially
    std::unique_lock<std::mutex> region(region_mu_, std::try_to_lock);
    if (workers_.empty() || n <= grain || !region.owns_lock()) {
      for (int i = 0; i < n; i++) fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      fn_ = &fn;
      n_ = n;
      grain_ = grain;
      next_.store(0);
      pending_ = (int)workers_.size();
      ++epoch_;
    }
    cv_.notify_all();
    run_chunks();
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  ThreadPool() {
    int want = 0;
    if (const char* e = std::getenv("SFMX_HOST_THREADS")) want = std::atoi(e);
    if (want <= 0) want = std::min(8, std::max(1, (int)std::thread::hardware_concurrency() / 2));
    for (int i = 1; i < want; i++) workers_.emplace_back([this] { worker(); });
  }
  ~ThreadPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
      ++epoch_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void run_chunks() {
    for (;;) {
      const int b = next_.fetch_add(grain_);
      if (b >= n_) break;
      const int e = std::min(n_, b + grain_);
      for (int i = b; i < e; i++) (*fn_)(i);
    }
  }
  void worker() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
      }
      run_chunks();
      {
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_, region_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, grain_ = 16, pending_ = 0;
  std::atomic<int> next_{0};
  unsigned long epoch_ = 0;
  bool stop_ = false;
};

}  // namespace sfmx_host
