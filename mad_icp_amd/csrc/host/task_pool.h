// A small persistent task pool for the host tree builder.  Forking a thread per task (std::async, what the reference
// does, mad_tree.cpp:104-111) costs tens of microseconds each — as much as the work of a task near the bottom of the
// forked levels — and an OpenMP-task version was measured slower still next to other OpenMP users in the process.
// Waiting HELPS: a thread that waits for a job runs queued jobs meanwhile, so nested fork/join cannot deadlock, and
// the pool keeps working (on the waiting thread alone) even if its workers are gone (e.g. in a forked child).
// Idle threads POLL for ~100 us before they sleep on the condition variable: a tree build forks a dozen times within two
// milliseconds, and a wake-up through the condition variable (tens of microseconds, a thundering herd with 31 workers)
// at every fork was a quarter of the build (measured: cutting the root's bounding-box pass over the pool made the build
// 1 ms SLOWER).  Between scans everybody sleeps.
#pragma once
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    bool wake;
    {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(j);
      queued_.store(static_cast<int>(q_.size()), std::memory_order_release);
      wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_all();  // (only threads that went to sleep need it: a worker that ran a task within the
                                 // last ~100 us is still polling `queued_`)
    return j;
  }

  void wait(const Handle& j) {
    for (;;) {
      if (j->done.load(std::memory_order_acquire)) break;
      if (Handle other = try_pop(1 << 30)) {  // waiting HELPS
        run(other);
        continue;
      }
      // nothing queued: the job is running on another thread.  Poll for a while — a fork's join is usually tens of
      // microseconds away, a condition-variable round trip costs more than that — then sleep.
      bool seen = false;
      for (int s = 0; s < kJoinSpins; ++s) {
        if (j->done.load(std::memory_order_acquire) || queued_.load(std::memory_order_acquire) > 0) {
          seen = true;
          break;
        }
        cpu_relax();
      }
      if (seen) continue;
      std::unique_lock<std::mutex> lk(m_);
      while (!j->done.load(std::memory_order_acquire) && q_.empty()) {
        ++sleepers_;
        cv_.wait(lk);
        --sleepers_;
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
      limit_.store(std::max(1, n), std::memory_order_release);
    }
    cv_.notify_all();
    cv_parked_.notify_all();
  }

  // several jobs at once: one lock, one wake-up (a build ends with a dozen 10-microsecond copy tasks: submitted one by
  // one, each paid its own notify)
  std::vector<Handle> submit_batch(std::vector<std::function<void()>> fns) {
    std::vector<Handle> out;
    out.reserve(fns.size());
    for (auto& f : fns) {
      Handle j = std::make_shared<Job>();
      j->fn = std::move(f);
      out.push_back(std::move(j));
    }
    bool wake;
    {
      std::lock_guard<std::mutex> lk(m_);
      for (const Handle& j : out) q_.push_back(j);
      queued_.store(static_cast<int>(q_.size()), std::memory_order_release);
      wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_all();
    return out;
  }

 private:
  // how long an idle thread polls before it sleeps: ~100 us — longer than the gap between two forks of one tree build,
  // far shorter than the gap between two scans
  static constexpr int kSpins = 4000;
  // a JOIN polls longer (~1 ms): the joins of a build form its critical path — one thread per fork level — and a wake-up
  // through the condition variable at each of six levels was ~0.1 ms of a 2 ms build
  static constexpr int kJoinSpins = 40000;
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }

  // Where the workers may run.  A tree build hands ranges of ONE 3 MB array from thread to thread at every fork; on a
  // two-socket, sixteen-L3 host (the GPU box: 2 x EPYC 9575F) unpinned workers land on other sockets and every hand-over
  // becomes cross-socket cache traffic.  So the workers are confined to the neighbourhood of the thread that creates
  // the pool: its NUMA node by default, its L3 domain with MADICP_HOST_AFFINITY=l3, anywhere with =none.  Only the
  // pool's own threads are touched (never the caller's), and only within the process's current affinity mask.
  static bool read_cpulist(const char* path, cpu_set_t* out) {
    std::FILE* f = std::fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!ok) return false;
    CPU_ZERO(out);
    int any = 0;
    for (char* p = buf; *p && *p != '\n';) {
      char* end = nullptr;
      const long a = std::strtol(p, &end, 10);
      if (end == p) break;
      long b = a;
      p = end;
      if (*p == '-') {
        b = std::strtol(p + 1, &end, 10);
        p = end;
      }
      for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
        CPU_SET(static_cast<int>(c), out);
        ++any;
      }
      if (*p == ',') ++p;
    }
    return any > 0;
  }
  static bool neighbourhood(cpu_set_t* out) {
    const char* how = std::getenv("MADICP_HOST_AFFINITY");
    if (how && std::strcmp(how, "none") == 0) return false;
    const int cpu = sched_getcpu();
    if (cpu < 0) return false;
    char path[256];
    cpu_set_t want;
    bool ok = false;
    if (how && std::strcmp(how, "l3") == 0) {
      std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
      ok = read_cpulist(path, &want);
    } else {
      for (int node = 0; node < 64 && !ok; ++node) {
        std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/node%d/cpulist", cpu, node);
        ok = read_cpulist(path, &want);
      }
    }
    if (!ok) return false;
    cpu_set_t have;
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return false;
    CPU_AND(out, &want, &have);
    return CPU_COUNT(out) >= 2;
  }

  TaskPool() {
    const int hw = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
    // never more workers than CPUs this process may run on (a container's mask, taskset): workers poll while idle
    cpu_set_t allowed;
    const int usable = sched_getaffinity(0, sizeof(allowed), &allowed) == 0 ? std::max(1, CPU_COUNT(&allowed)) : hw;
    const int n = std::max(0, std::min(std::min(hw, usable), 32) - 1);
    cpu_set_t where;
    // pinned to the creating thread's NUMA node / L3 neighbourhood only when that neighbourhood has room for the pool and its
    // caller: on NPS4 parts or small masks the workers would otherwise queue on a handful of CPUs (MADICP_HOST_AFFINITY=none
    // switches pinning off altogether)
    const bool pin = neighbourhood(&where) && CPU_COUNT(&where) >= n + 1;
    threads_.reserve(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
      threads_.emplace_back([this, i] { worker(i); });
      if (pin) pthread_setaffinity_np(threads_.back().native_handle(), sizeof(where), &where);
    }
    for (std::thread& t : threads_) t.detach();
  }

  // the oldest queued job, if worker `index` may take one under the current limit (the waiting caller: always)
  Handle try_pop(int index_budget_slack) {
    if (queued_.load(std::memory_order_acquire) <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(m_);
    if (q_.empty() || index_budget_slack <= 0) return nullptr;
    Handle j = std::move(q_.front());
    q_.pop_front();
    queued_.store(static_cast<int>(q_.size()), std::memory_order_release);
    return j;
  }

  void run(const Handle& j) {
    try {
      j->fn();
    } catch (...) {
      j->error = std::current_exception();  // a detached worker must not terminate the process; the waiter rethrows
    }
    j->done.store(true, std::memory_order_release);
    bool wake;
    {
      std::lock_guard<std::mutex> lk(m_);  // pairs with a waiter's check-then-sleep under the same mutex
      wake = sleepers_ > 0;
    }
    if (wake) cv_.notify_all();
  }

  void worker(int index) {
    for (;;) {
      const bool allowed = index < limit_.load(std::memory_order_acquire) - 1;
      if (allowed) {
        if (Handle j = try_pop(1)) {
          run(j);
          continue;
        }
        bool seen = false;
        for (int s = 0; s < kSpins; ++s) {
          if (queued_.load(std::memory_order_acquire) > 0) {
            seen = true;
            break;
          }
          cpu_relax();
        }
        if (seen) continue;
      }
      std::unique_lock<std::mutex> lk(m_);
      if (index >= limit_.load(std::memory_order_relaxed) - 1) {
        // over the caller's thread budget: parked on a condition variable of its own, which only set_limit() signals —
        // on the queue's one, every submit would wake these threads just to send them back to sleep (with a budget of
        // 16 on a pool of 31 that thundering herd was ~14 us per submit, 0.2 ms at the end of every build)
        cv_parked_.wait(lk, [this, index] { return index < limit_.load(std::memory_order_relaxed) - 1; });
        continue;
      }
      if (!q_.empty()) continue;
      ++sleepers_;
      cv_.wait(lk);
      --sleepers_;
    }
  }

  std::mutex m_;
  std::condition_variable cv_;
  std::condition_variable cv_parked_;  // workers beyond the thread budget
  std::deque<Handle> q_;
  std::atomic<int> queued_{0};  // q_.size(), readable without the mutex
  int sleepers_ = 0;            // threads inside cv_.wait (under m_)
  std::vector<std::thread> threads_;
  std::atomic<int> limit_{1 << 30};
};

}  // namespace madicp_host
