// pool.h -- a small fork-join thread pool for the host-side per-agent loops (environment stepping in sim.cpp,
// exploration noise in agents.hip).  Work items are independent; the result does not depend on the thread count.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// fork-join over [0, n): the calling thread takes part; workers sleep between calls
class DdrlPool {
 public:
  explicit DdrlPool(int threads) {
    for (int t = 1; t < threads; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~DdrlPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; ++epoch_; }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }
  void run(int n, const std::function<void(int)> &f) {
    if (workers_.empty() || n < 2) { for (int i = 0; i < n; ++i) f(i); return; }
    { std::lock_guard<std::mutex> l(m_); f_ = &f; n_ = n; next_ = 0; pending_ = (int)workers_.size(); ++epoch_; }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return pending_ == 0; });
    f_ = nullptr;
  }

 private:
  void work() { for (int i; (i = next_.fetch_add(1)) < n_;) (*f_)(i); }
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
      }
      work();
      { std::lock_guard<std::mutex> l(m_); --pending_; }
      done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)> *f_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, pending_ = 0;
  unsigned long epoch_ = 0;
  bool stop_ = false;
};

