// hostpar.h — host-side plumbing of libdada2hip.so that is not the algorithm: a small worker pool for the
// input marshalling of the boundary call (the reference copies its R inputs serially, Rmain.cpp:102-120; at
// 10^6 uniques that is 2 GB of doubles and 250 MB of characters, so here it is spread over host threads), and
// caches of device / pinned allocations so that back-to-back boundary calls do not pay hipMalloc / hipFree /
// hipHostMalloc again ("no global state except an optional per-device context cache").
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "knobs.h"

#include <unistd.h>

#include <hip/hip_runtime.h>

namespace d2 {

// ---- worker pool ---------------------------------------------------------------------------------
// Workers are spawned once per PROCESS: a fork()ed child (Python multiprocessing's default start method, R's
// parallel::mclapply) inherits the pool object but none of its threads, so the pool remembers the pid it spawned in
// and rebuilds itself in a child.  A piece that throws stops the job and the exception is rethrown in the caller;
// a parallel_for issued from inside a piece runs inline.
class HostPool {
 public:
  static HostPool &get() {
    static HostPool *p = new HostPool();   // leaked on purpose: no join at static destruction (a forked child has no threads to join)
    return *p;
  }
  int nthreads() const { return nthreads_; }
  // f(lo, hi) over [0, n) in pieces of `grain`; the caller takes part, returns when all pieces are done.
  // max_threads > 0: at most that many threads take pieces (the others wake, see that the job is not theirs and go back to sleep)
  void run(size_t n, size_t grain, const std::function<void(size_t, size_t)> &f, int max_threads = 0) {
    if (n == 0) return;
    if (grain == 0) grain = 1;
    if (nthreads_ <= 1 || n <= grain || in_job()) { f(0, n); return; }
    if (getpid() != pid_) respawn_after_fork();
    std::unique_lock<std::mutex> job_lock(S->job_mu);   // one job at a time
    {
      std::lock_guard<std::mutex> g(S->mu);
      fn_ = &f; n_ = n; grain_ = grain; next_.store(0); pending_ = (int)S->workers.size(); failed_.store(false); exc_ = nullptr;
      cap_ = max_threads > 0 ? max_threads : nthreads_;
      gen_++;
    }
    S->cv.notify_all();
    work();
    {
      std::unique_lock<std::mutex> l(S->mu);
      S->done_cv.wait(l, [&] { return pending_ == 0; });
      fn_ = nullptr;
    }
    if (exc_) { std::exception_ptr e = exc_; exc_ = nullptr; std::rethrow_exception(e); }
  }

 private:
  // everything that cannot survive a fork lives behind one pointer and is simply abandoned in the child
  struct Sync {
    std::vector<std::thread> workers;
    std::mutex mu, job_mu;
    std::condition_variable cv, done_cv;
  };
  HostPool() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    if (n > 64) n = 64;                                  // memory-bound copies: 16 / 32 / 64 threads = 55 / 34 / 25 ms for the 1M-unique upload (profiles/r02q_bench_cfg3_threads*.json)
    {
      // A container's CPU quota (cgroup v2 cpu.max: "<quota us> <period us>", "max" = none) is invisible to hardware_concurrency():
      // the GPU boxes show 256 cores and allow sixteen CPUs' worth per 100 ms.  The marshalling is a burst (64 threads x 25 ms =
      // one whole period's quota at 16 CPUs, measured best there: profiles/r10s_*), so the pool may be four times the quota
      // wide and no wider - a 4-CPU container gets 16 threads instead of 64 that the kernel would park most of the time.
      if (FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = "";
        long long period = 0;
        if (fscanf(fp, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
          const long long cpus = (atoll(q) + period - 1) / period;
          if (cpus > 0 && 4 * cpus < n) n = (int)std::max<long long>(2, 4 * cpus);
        }
        fclose(fp);
      }
    }
    if (knobs().host_threads > 0) n = knobs().host_threads;
    nthreads_ = n;
    spawn();
  }
  void spawn() {
    S = new Sync();
    pid_ = getpid();
    stop_ = false; pending_ = 0; fn_ = nullptr;
    // workers start from the generation of their birth: a respawned pool must not replay the parent's last job
    for (int i = 1; i < nthreads_; i++) S->workers.emplace_back([this, g0 = gen_, i] { loop(g0, i); });
  }
  void respawn_after_fork() {
    static std::mutex fork_mu;
    std::lock_guard<std::mutex> g(fork_mu);
    if (getpid() == pid_) return;
    // the parent's Sync (thread handles of threads this process never had, mutexes in whatever state they were forked
    // in) is leaked, not destroyed: ~thread on a joinable handle would terminate the process
    spawn();
  }
  static bool &in_job() { static thread_local bool v = false; return v; }
  void work() {
    in_job() = true;
    for (;;) {
      const size_t lo = next_.fetch_add(grain_);
      if (lo >= n_) break;
      if (failed_.load(std::memory_order_relaxed)) continue;   // drain the remaining pieces without running them
      try {
        (*fn_)(lo, std::min(n_, lo + grain_));
      } catch (...) {
        std::lock_guard<std::mutex> g(S->mu);
        if (!exc_) exc_ = std::current_exception();
        failed_.store(true);
      }
    }
    in_job() = false;
  }
  void loop(uint64_t seen, int index) {
    Sync *my = S;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(my->mu);
        my->cv.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      if (index < cap_) work();                        // (the caller is thread 0 of the job)
      {
        std::lock_guard<std::mutex> g(my->mu);
        if (--pending_ == 0) my->done_cv.notify_all();
      }
    }
  }
  int nthreads_ = 1;
  int cap_ = 1;                  // threads that take pieces of the job in hand (run's max_threads)
  pid_t pid_ = 0;
  Sync *S = nullptr;
  const std::function<void(size_t, size_t)> *fn_ = nullptr;
  size_t n_ = 0, grain_ = 1;
  std::atomic<size_t> next_{0};
  std::atomic<bool> failed_{false};
  std::exception_ptr exc_ = nullptr;
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

inline void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &f, int max_threads = 0) { HostPool::get().run(n, grain, f, max_threads); }

// ---- allocation caches ---------------------------------------------------------------------------
// Size classes: multiples of 1/4 of the power of two below the request (<= 25 % slack), at least 256 B; a freed block
// goes back to its class and is handed out again to the next request of the same class on the same device.
inline size_t alloc_class(size_t bytes) {
  if (bytes < 256) return 256;
  size_t p = 256;
  while ((p << 1) <= bytes) p <<= 1;
  const size_t step = p >> 2;
  return (bytes + step - 1) / step * step;
}

class AllocCache {
 public:
  static AllocCache &get() {
    static AllocCache *c = new AllocCache();   // leaked on purpose: the HIP runtime may be gone at static destruction
    return *c;
  }
  hipError_t dev_alloc(void **p, size_t bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t cls = alloc_class(bytes);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto &fl = dev_free_[dev];
      auto it = fl.find(cls);
      if (it != fl.end()) {
        *p = it->second;
        fl.erase(it);
        dev_cached_ -= cls;
        dev_live_[*p] = {dev, cls};
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {   // out of memory: give the cache back and retry once
      trim();
      (void)hipGetLastError();
      e = hipMalloc(p, cls);
      if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(mu_);
    dev_live_[*p] = {dev, cls};
    return hipSuccess;
  }
  void dev_release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = dev_live_.find(p);
    if (it == dev_live_.end()) { (void)hipFree(p); return; }
    const Blk b = it->second;
    dev_live_.erase(it);
    // unwinding from an error (a timed-out wait, a failed launch): kernels that use this block may still be running, so it
    // must not be handed to another call - hipFree synchronises the device first, the cache would not
    if (std::uncaught_exceptions() > 0) { (void)hipDeviceSynchronize(); (void)hipFree(p); return; }
    if (!enabled_ || dev_cached_ + b.cls > dev_cap()) { (void)hipFree(p); return; }
    dev_free_[b.dev].emplace(b.cls, p);
    dev_cached_ += b.cls;
  }
  hipError_t pin_alloc(void **p, size_t bytes) {
    const size_t cls = alloc_class(bytes);
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = pin_free_.find(cls);
      if (it != pin_free_.end()) {
        *p = it->second;
        pin_free_.erase(it);
        pin_cached_ -= cls;
        pin_live_[*p] = cls;
        return hipSuccess;
      }
    }
    hipError_t e = hipHostMalloc(p, cls, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu_);
    pin_live_[*p] = cls;
    return hipSuccess;
  }
  void pin_release(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> g(mu_);
    auto it = pin_live_.find(p);
    if (it == pin_live_.end()) { (void)hipHostFree(p); return; }
    const size_t cls = it->second;
    pin_live_.erase(it);
    if (!enabled_ || pin_cached_ + cls > pin_cap_) { (void)hipHostFree(p); return; }
    pin_free_.emplace(cls, p);
    pin_cached_ += cls;
  }
  // hand every cached block back to the runtime (dada2hip_trim_cache)
  void trim() {
    std::lock_guard<std::mutex> g(mu_);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &d : dev_free_) {
      (void)hipSetDevice(d.first);
      for (auto &b : d.second) (void)hipFree(b.second);
      d.second.clear();
    }
    (void)hipSetDevice(cur);
    dev_cached_ = 0;
    for (auto &b : pin_free_) (void)hipHostFree(b.second);
    pin_free_.clear();
    pin_cached_ = 0;
  }
  size_t cached_bytes() {
    std::lock_guard<std::mutex> g(mu_);
    return dev_cached_ + pin_cached_;
  }

 private:
  AllocCache() {
    if (knobs().alloc_cache >= 0) enabled_ = knobs().alloc_cache != 0;
    if (knobs().alloc_cache_gb >= 0) cap_ = (size_t)knobs().alloc_cache_gb << 30;
  }
  struct Blk { int dev; size_t cls; };
  // cached device bytes are capped at a third of the device's memory (DADA2HIP_ALLOC_CACHE_GB overrides)
  size_t dev_cap() {
    if (cap_ == 0) {
      size_t tot = 0;   // (hipDeviceTotalMem: hipMemGetInfo is refused while another host thread captures a graph)
      int dev = 0;
      (void)hipGetDevice(&dev);
      cap_ = (hipDeviceTotalMem(&tot, dev) == hipSuccess && tot) ? tot / 3 : ((size_t)32 << 30);
    }
    return cap_;
  }
  std::mutex mu_;
  std::map<int, std::multimap<size_t, void *>> dev_free_;
  std::map<void *, Blk> dev_live_;
  std::multimap<size_t, void *> pin_free_;
  std::map<void *, size_t> pin_live_;
  size_t dev_cached_ = 0, pin_cached_ = 0;
  size_t cap_ = 0, pin_cap_ = (size_t)8 << 30;   // cap_ 0 = not yet derived from the device (dev_cap)
  bool enabled_ = true;
};

}  // namespace d2
