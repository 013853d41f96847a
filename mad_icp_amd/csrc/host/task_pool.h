// A small persistent task pool for the host tree builder.  Forking a thread per task (std::async, what the reference
// does, mad_tree.cpp:104-111) costs tens of microseconds each — as much as the work of a task near the bottom of the
// forked levels — and an OpenMP-task version was measured slower still next to other OpenMP users in the process.
// Waiting HELPS: a thread that waits for a job runs queued jobs meanwhile, so nested fork/join cannot deadlock, and
// the pool keeps working (on the waiting thread alone) even if its workers are gone (e.g. in a forked child).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace madicp_host {

class TaskPool {
 public:
  struct Job {
    std::function<void()> fn;
    std::atomic<bool> done{false};
    std::exception_ptr error;  // what fn threw, rethrown by wait()
  };
  using Handle = std::shared_ptr<Job>;

  static TaskPool& instance() {
    static TaskPool* pool = new TaskPool();  // never destroyed: idle workers die with the process
    return *pool;
  }

  Handle submit(std::function<void()> fn) {
    Handle j = std::make_shared<Job>();
    j->fn = std::move(fn);
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(j);
    }
    cv_.notify_all();
    return j;
  }

  void wait(const Handle& j) {
    std::unique_lock<std::mutex> lk(m_);
    while (!j->done.load(std::memory_order_acquire)) {
      if (!q_.empty()) {
        Handle other = std::move(q_.front());
        q_.pop_front();
        lk.unlock();
        run(other);
        lk.lock();
      } else {
        cv_.wait(lk);
      }
    }
    if (j->error) std::rethrow_exception(j->error);
  }

  int workers() const { return static_cast<int>(threads_.size()); }

  // The caller's thread budget (the reference caps OpenMP and its async forks at num_threads: pipeline.cpp:64-65):
  // at most `n` threads — the caller included — run tasks at any time; the other workers sleep.  Process-wide, like
  // omp_set_num_threads; the last caller wins.
  void set_limit(int n) {
    {
      std::lock_guard<std::mutex> lk(m_);
      limit_ = std::max(1, n);
    }
    cv_.notify_all();
  }

 private:
  TaskPool() {
    const int hw = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
    const int n = std::max(0, std::min(hw, 32) - 1);
    threads_.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) threads_.emplace_back([this, i] { worker(i); });
    for (std::thread& t : threads_) t.detach();
  }

  void run(const Handle& j) {
    try {
      j->fn();
    } catch (...) {
      j->error = std::current_exception();  // a detached worker must not terminate the process; the waiter rethrows
    }
    {
      std::lock_guard<std::mutex> lk(m_);  // pairs with the waiter's check under the same mutex
      j->done.store(true, std::memory_order_release);
    }
    cv_.notify_all();
  }

  void worker(int index) {
    std::unique_lock<std::mutex> lk(m_);
    for (;;) {
      cv_.wait(lk, [this, index] { return !q_.empty() && index < limit_ - 1; });
      Handle j = std::move(q_.front());
      q_.pop_front();
      lk.unlock();
      run(j);
      lk.lock();
    }
  }

  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Handle> q_;
  std::vector<std::thread> threads_;
  int limit_ = 1 << 30;
};

}  // namespace madicp_host
