// host_pipeline.hpp -- staging of HOST-pointer batches (the reference's own calling convention, src/main/femto.c:275)
// onto the GPU: a small persistent worker pool copies/validates the caller's pageable arrays into pinned chunks
// while the previous chunk travels over PCIe and the one before is searched.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace femto_amd {

class WorkerPool {
 public:
  explicit WorkerPool(int n) : n_(n < 1 ? 1 : n) {
    for (int t = 1; t < n_; t++) threads_.emplace_back([this, t] { loop(t); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }
  int size() const { return n_; }
  // runs fn(t, n) for t in [0, n) on the pool (the caller is worker 0) and returns when all are done
  void run(const std::function<void(int, int)>& fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    fn(0, n_);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void loop(int t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(int, int)>* fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
      }
      if (fn) (*fn)(t, n_);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int, int)>* fn_ = nullptr;
  uint64_t gen_ = 0;
  int pending_ = 0;
  bool stop_ = false;
};

}  // namespace femto_amd
