// host_pipeline.hpp -- staging of HOST-pointer batches (the reference's own calling convention, src/main/femto.c:275)
// onto the GPU: a small persistent worker pool copies/validates the caller's pageable arrays into pinned chunks
// while the previous chunk travels over PCIe and the one before is searched.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace femto_amd {

// A batch call issues a dozen short jobs (0.1 - 0.5 ms each) back to back.  Waking 127 sleeping threads through a
// condition variable costs about as much as such a job (every waiter re-takes the mutex in turn), so the workers SPIN on
// the job generation for a short while after a job (kSpinUs: longer than the gap between two jobs of one call) and only
// then go to sleep; the caller spins on the completion count.  Sleeping workers are woken through the condition
// variable; `sleepers_` (seq_cst against `gen_`) decides whether a notify is needed.
class WorkerPool {
 public:
  explicit WorkerPool(int n) : n_(n < 1 ? 1 : n) {
    for (int t = 1; t < n_; t++) threads_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    stop_.store(true);
    gen_.fetch_add(1);
    {
      std::lock_guard<std::mutex> lk(mu_);
      cv_.notify_all();
    }
    for (auto& th : threads_) th.join();
  }
  int size() const { return n_; }
  // runs fn(t, n) for t in [0, n) on the pool (the caller is worker 0) and returns when all are done;
  // one job at a time (callers serialise)
  void run(const std::function<void(int, int)>& fn) {
    fn_ = &fn;
    pending_.store(n_ - 1, std::memory_order_relaxed);
    gen_.fetch_add(1);                       // publishes fn_ and pending_
    if (sleepers_.load() > 0) {
      std::lock_guard<std::mutex> lk(mu_);
      cv_.notify_all();
    }
    fn(0, n_);
    int spins = 0;
    while (pending_.load(std::memory_order_acquire) != 0) {
      cpu_relax();
      if (++spins > (1 << 14)) {             // a worker was descheduled: stop burning its core
        std::this_thread::yield();
        spins = 0;
      }
    }
    fn_ = nullptr;
  }

 private:
  static constexpr int kSpinUs = 400;
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      // spin, then sleep
      bool fresh = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0;; i++) {
        if (gen_.load(std::memory_order_acquire) != seen) { fresh = true; break; }
        cpu_relax();
        if ((i & 255) == 255 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > kSpinUs)
          break;
      }
      if (!fresh) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return gen_.load() != seen; });
        sleepers_.fetch_sub(1);
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load()) return;
      const std::function<void(int, int)>* fn = fn_;
      if (fn) (*fn)(t, n_);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  const std::function<void(int, int)>* fn_ = nullptr;
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> pending_{0};
  std::atomic<int> sleepers_{0};
  std::atomic<bool> stop_{false};
};

}  // namespace femto_amd
