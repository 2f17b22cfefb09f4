// Worker threads shared by the host-side parallel regions of the library (hostio.cpp, snps_abi.hip).
#pragma once
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <exception>
#include <functional>
#include <sched.h>
#include <mutex>
#include <thread>
#include <vector>

namespace midas {

// CPUs this process may actually use: the hardware threads, or fewer when a cgroup CPU quota (cpu.max of cgroup v2,
// cpu.cfs_quota_us / cpu.cfs_period_us of v1) says so.  A container that shows 256 hardware threads under a quota of
// 16 CPUs runs 128 busy threads for an eighth of every 100 ms period and is throttled for the rest of it: more threads
// than the quota buy nothing and turn every wait into a stall of most of a period.  Under torchrun the budget is the
// rank's share of the node (LOCAL_WORLD_SIZE ranks run side by side).
inline int cpu_budget() {
  static const int budget = [] {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    {     // the CPUs the process may run on (taskset, numactl, a Slurm cpuset without a quota): never more threads than those
      cpu_set_t set;
      CPU_ZERO(&set);
      if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int n = CPU_COUNT(&set);
        if (n >= 1 && (unsigned)n < hw) hw = (unsigned)n;
      }
    }
    double quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0};
      double p = 0;
      if (fscanf(f, "%63s %lf", q, &p) == 2 && q[0] != 'm' && p > 0) { quota = atof(q); period = p; }
      fclose(f);
    } else {
      FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
      FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
      if (fq && fp && fscanf(fq, "%lf", &quota) == 1 && fscanf(fp, "%lf", &period) == 1) { /* read */ } else { quota = -1; }
      if (fq) fclose(fq);
      if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) {
      const unsigned cap = (unsigned)((quota + period - 1) / period);
      if (cap >= 1 && cap < hw) hw = cap;
    }
    // one process per GPU under torchrun: the node's CPUs are shared by LOCAL_WORLD_SIZE ranks
    if (const char* lws = getenv("LOCAL_WORLD_SIZE")) {
      char* end = nullptr;
      const long ranks = strtol(lws, &end, 10);
      if (end != lws && *end == '\0' && ranks > 1 && ranks <= 4096) hw = hw / (unsigned)ranks > 0 ? hw / (unsigned)ranks : 1u;
    }
    return (int)hw;
  }();
  return budget;
}

// Waiting for another thread's progress without spinning (a yield loop burns the CPU quota of everybody else in the
// cgroup): the waiter sleeps on a condition variable until the predicate holds; whoever changes what a predicate reads
// calls signal() afterwards.
class Events {
 public:
  template <class P>
  void wait(P&& pred) {
    if (pred()) return;
    std::unique_lock<std::mutex> g(m_);
    cv_.wait(g, pred);
  }
  void signal() {
    { std::lock_guard<std::mutex> g(m_); }      // (a waiter is either before its check or already asleep)
    cv_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
};

// The library's worker threads.  Every parallel region of the host code (inflate, record decode, row formatting + gzip, table
// parsing) used to start its own std::threads and join them: on a 256-thread host that is ~130 thread starts per region,
// several regions per species.  Here the threads are started once and parked on a condition variable between regions.
// One region runs at a time; a caller that finds the pool taken (another Python thread inside the library) starts
// threads of its own, as before.  The pool is never torn down (the parked threads end with the process) and is rebuilt in
// a forked child, where the parent's threads do not exist.
class Workers {
 public:
  // work() runs on up to nt threads (the caller is one of them) and must be written as "claim the next item until none
  // is left": the region ends when the caller's own work() has returned and every pool thread that started on it has
  // returned too.  A pool thread the host was slow to wake (a busy machine can take tens of milliseconds) finds the
  // region closed and goes back to sleep -- nobody waits for a thread that has nothing left to do.
  // An exception thrown by work() on a pool thread or on a helper thread is caught there and rethrown by run() on the
  // caller once the region is closed (the first one wins): no std::terminate, no job pointer left dangling, the pool
  // never stays marked as taken.
  static void run(int nt, const std::function<void()>& work) {
    if (nt <= 1) { work(); return; }
    std::exception_ptr failed;
    std::mutex failed_m;
    const std::function<void()> guarded = [&] {
      try {
        work();
      } catch (...) {
        std::lock_guard<std::mutex> g(failed_m);
        if (!failed) failed = std::current_exception();
      }
    };
    Workers* w = instance();
    if (w->taken_.exchange(true, std::memory_order_acquire)) {     // another region is running (or this is a nested one)
      std::vector<std::thread> th;
      try {
        for (int t = 1; t < nt; ++t) th.emplace_back(guarded);
      } catch (...) {                                              // (thread creation failed: go on with the ones we have)
      }
      guarded();
      for (auto& x : th) x.join();
    } else {
      struct Region {                                              // closes the region and frees the pool on every way out
        Workers* w;
        ~Region() { w->close(); w->taken_.store(false, std::memory_order_release); }
      } region{w};
      w->open(nt - 1, &guarded);
      guarded();
    }
    if (failed) std::rethrow_exception(failed);
  }

 private:
  static Workers* instance() {
    // (no lock: a mutex here could be held by another thread at fork() and would stay locked in the child for ever.  Two
    // threads racing to make the first pool leak one empty object; the pool of a forked child is made anew, the parent's
    // -- whose threads do not exist in the child -- is leaked.)
    static std::atomic<Workers*> self{nullptr};
    Workers* w = self.load(std::memory_order_acquire);
    if (!w || w->pid_ != getpid()) {
      Workers* made = new Workers();
      if (self.compare_exchange_strong(w, made, std::memory_order_acq_rel)) w = made;
      else if (w && w->pid_ == getpid()) delete made;
      else { self.store(made, std::memory_order_release); w = made; }
    }
    return w;
  }
  Workers() : pid_(getpid()) {}
  void open(int n, const std::function<void()>* job) {
    std::unique_lock<std::mutex> g(m_);
    while ((int)th_.size() < n) {
      const int id = (int)th_.size();
      th_.emplace_back([this, id] { loop(id); });
      th_.back().detach();
    }
    job_ = job;
    want_ = n;
    inside_ = 0;
    is_open_ = true;
    ++gen_;
    g.unlock();
    wake_.notify_all();
  }
  void close() {
    std::unique_lock<std::mutex> g(m_);
    is_open_ = false;                                   // late wakers stay out from here on
    done_.wait(g, [this] { return inside_ == 0; });
    job_ = nullptr;
  }
  void loop(int id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void()>* job;
      {
        std::unique_lock<std::mutex> g(m_);
        wake_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (id >= want_ || !is_open_) continue;
        job = job_;
        ++inside_;
      }
      (*job)();
      std::lock_guard<std::mutex> g(m_);
      if (--inside_ == 0) done_.notify_one();
    }
  }
  const pid_t pid_;
  std::atomic<bool> taken_{false};
  std::mutex m_;
  std::condition_variable wake_, done_;
  std::vector<std::thread> th_;
  const std::function<void()>* job_ = nullptr;
  uint64_t gen_ = 0;
  int want_ = 0, inside_ = 0;
  bool is_open_ = false;
};

}  // namespace midas
