// Shared-memory helpers with the reference's names (src/base/parallel.hpp, parallel/parallel_lambda.hpp,
// parallel/thread_pool.hpp): --num_thread, parallel_for (static blocks), dynamic_parallel_for (work
// queue), ThreadPool.  Used by the evaluation loop; CDAE training itself runs on the GPU.
#ifndef CDAE_HOST_BASE_PARALLEL_HPP_
#define CDAE_HOST_BASE_PARALLEL_HPP_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include <gflags/gflags.h>

DEFINE_int32(num_thread, 1, "NUM OF THREADS");

namespace libcf {

inline size_t num_hardware_threads() {
  if (FLAGS_num_thread > 0) return static_cast<size_t>(FLAGS_num_thread);
  const unsigned hc = std::thread::hardware_concurrency();
  return hc ? hc : 1;
}

// fn(thread_index, num_threads) on num_hardware_threads() threads
inline void in_parallel(const std::function<void(size_t, size_t)>& fn) {
  const size_t n = num_hardware_threads();
  if (n <= 1) { fn(0, 1); return; }
  std::vector<std::thread> pool;
  pool.reserve(n);
  for (size_t t = 0; t < n; ++t) pool.emplace_back(fn, t, n);
  for (auto& th : pool) th.join();
}

inline void parallel_for(size_t first, size_t last, const std::function<void(size_t)>& fn) {
  in_parallel([&](size_t tid, size_t nt) {
    const size_t len = last - first, per = (len + nt - 1) / nt;
    const size_t a = first + tid * per, b = a + per < last ? a + per : last;
    for (size_t i = a; i < b; ++i) fn(i);
  });
}

template <class It>
inline void parallel_for_each(const It& first, const It& last, const std::function<void(decltype(*first)&)>& fn) {
  const size_t len = static_cast<size_t>(last - first);
  parallel_for(0, len, [&](size_t i) { fn(*(first + i)); });
}

class ThreadPool {
 public:
  explicit ThreadPool(size_t n) {
    for (size_t i = 0; i < (n ? n : 1); ++i)
      workers_.emplace_back([this] {
        for (;;) {
          std::function<void()> job;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
            if (stop_ && jobs_.empty()) return;
            job = std::move(jobs_.front());
            jobs_.pop();
          }
          job();
        }
      });
  }
  template <class F>
  std::future<void> enqueue(F&& f) {
    auto task = std::make_shared<std::packaged_task<void()>>(std::forward<F>(f));
    std::future<void> fut = task->get_future();
    { std::lock_guard<std::mutex> lk(mu_); jobs_.emplace([task] { (*task)(); }); }
    cv_.notify_one();
    return fut;
  }
  ~ThreadPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
 private:
  std::vector<std::thread> workers_;
  std::queue<std::function<void()>> jobs_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
};

// indices handed out one at a time (load-balanced)
inline void dynamic_parallel_for(size_t first, size_t last, const std::function<void(size_t)>& fn) {
  std::atomic<size_t> next(first);
  in_parallel([&](size_t, size_t) {
    for (size_t i = next.fetch_add(1); i < last; i = next.fetch_add(1)) fn(i);
  });
}

}  // namespace libcf
#endif
