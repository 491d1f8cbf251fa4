// The library's persistent host worker pool (header-only: matcher_host.cpp instantiates it; tests/test_host_pool.py builds a
// stress program around it without a GPU).
#pragma once

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace kh
{

// ---- host worker pool ---------------------------------------------------------------------------
// The exact-arithmetic host half (tables, penalties, FindValidPoints, tie averaging, covariances) is
// O(P + nX*nY + nA) per match and independent between the matches of a batch; with the scoring kernel at
// ~30 us per match it is what bounds a batch, so it is spread over a few host threads
// (KH_HOST_THREADS, default min(32, cores): measured 16 / 32 / 64 -> 7.4 / 7.9 / 8.0 k loop-closure pairs/s and
// 61.6 / 61.9 / 59.4 k config-2 matches/s).  Results do not depend on the thread count.
class HostPool
{
public:
  static HostPool & instance() {static HostPool p(0); return p;}
  // the mapper's pose re-projections: milliseconds of uniform work over thousands of scans, kept alive between calls (a team
  // spawned per call cost a millisecond per loop closure).  64 threads: the work writes 35 KB per scan (points, filtered
  // points) and is bound by the host's memory bandwidth well before the box runs out of cores -- 192 threads took 7.9 s of
  // the 50 000-scan replay where 64 take 4
  static HostPool & wide() {static HostPool p(1); return p;}
  // runs fn(i) for i in [0, n); returns when all are done.  linger_ticks: how long the workers stay awake (spinning) BEHIND this
  // region before they sleep, where the caller knows that another region follows within that time and that its wake-up (~50 us: a
  // futex wake and the scheduler) would be exposed -- the two ends of a chunked batch call, csrc/matcher_host.cpp.  Never for
  // regions that follow one another all through a call: the boxes run under a CPU quota (see kSpinTicks).
  void run(size_t n, const std::function<void(size_t)> & fn, uint64_t linger_ticks = 0)
  {
    if (n == 0) {return;}
    // a region entered from inside a region (fn calling run() again, on the caller thread or on a worker) runs in line: the
    // flag, not a try_lock on a mutex this thread may already own (undefined behaviour)
    if (n == 1 || workers_.empty() || (inside_region() & bit_) != 0) {for (size_t i = 0; i < n; ++i) {fn(i);} return;}
    // one parallel region at a time; a second handle arriving from another thread while the workers are taken does its
    // loop itself instead of queueing behind the first (the regions are short: waiting would idle the caller's GPU stream)
    std::unique_lock<std::mutex> serial(run_mu_, std::try_to_lock);
    if (!serial.owns_lock()) {for (size_t i = 0; i < n; ++i) {fn(i);} return;}
    fn_ = &fn; n_ = n; next_.store(0, std::memory_order_relaxed);
    linger_.store(linger_ticks, std::memory_order_relaxed);
    failed_.store(false, std::memory_order_relaxed);
    pending_.store(static_cast<uint32_t>(workers_.size()), std::memory_order_relaxed);
    generation_.fetch_add(1, std::memory_order_release);
    futex_wake_all(&generation_);
    work();
    // every worker checks in (the last one wakes this thread): nobody can still be reading fn_ / n_ when run() returns --
    // also when fn threw: the first exception is kept and rethrown here, after the region has drained
    const uint64_t t0 = ticks();
    for (;;) {
      const uint32_t left = pending_.load(std::memory_order_acquire);
      if (left == 0) {break;}
      if (ticks() - t0 < kSpinTicks) {__builtin_ia32_pause();} else {futex_wait(&pending_, left);}
    }
    fn_ = nullptr;
    if (failed_.load(std::memory_order_acquire)) {
      std::exception_ptr e;
      std::swap(e, error_);
      std::rethrow_exception(e);
    }
  }
  ~HostPool()
  {
    stop_.store(true, std::memory_order_release);
    generation_.fetch_add(1, std::memory_order_release);
    futex_wake_all(&generation_);
    for (auto & t : workers_) {t.join();}
  }
private:
  // Sleeping and waking go through the futex of the generation counter itself: one system call wakes every worker and none
  // of them has a mutex to re-acquire on the way out (with a condition variable the 31 woken threads queued for its mutex one
  // after the other: 0.15 ms per region of 64 jobs measured in the chunk pipeline, most of it that queue).  A worker spins
  // for kSpinTicks (~25 us) before it sleeps: regions that follow one another directly (a chunk's finalisation, then the
  // next chunk's preparation) find the workers awake.  Never longer: a container with a CPU quota throttles a process whose
  // idle threads spin (measured: workers spinning for the length of a batch call halved the throughput of the bench).
  static constexpr uint64_t kSpinTicks = 60000;
public:
  static constexpr uint64_t kTicksPerMicrosecond = 2400;      // (nominal: the spin windows are not measurements)
private:
  static uint64_t ticks() {return __builtin_ia32_rdtsc();}
  static void futex_wait(std::atomic<uint32_t> * a, uint32_t expected)
  {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
  }
  static void futex_wake_all(std::atomic<uint32_t> * a)
  {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
  }
  explicit HostPool(int wide_pool) : bit_(wide_pool ? 2u : 1u)
  {
    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the futex word is the atomic itself");
    unsigned want = std::min(wide_pool ? 64u : 32u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char * e = std::getenv(wide_pool ? "KH_MAPPER_UPDATE_THREADS" : "KH_HOST_THREADS")) {want = static_cast<unsigned>(std::max(1, std::atoi(e)));}
    for (unsigned t = 1; t < want; ++t) {workers_.emplace_back([this] {loop();});}
  }
  // the pools whose regions this thread is running an item of, one bit per pool (per pool: a region of one pool may hand a loop to
  // the other; a region re-entered at any depth runs in line)
  static uint32_t & inside_region() {static thread_local uint32_t inside = 0; return inside;}
  const uint32_t bit_;
  void work()
  {
    const uint32_t outer = inside_region();
    inside_region() = outer | bit_;
    for (;;) {
      const size_t i = next_.fetch_add(1);
      if (i >= n_) {break;}
      try {
        (*fn_)(i);
      } catch (...) {
        // keep the first exception, drop the rest of the region's items
        if (!failed_.exchange(true, std::memory_order_acq_rel)) {error_ = std::current_exception();}
        next_.store(n_, std::memory_order_relaxed);
      }
    }
    inside_region() = outer;
  }
  void loop()
  {
    uint32_t seen = 0;
    uint64_t idle_since = ticks(), spin_for = kSpinTicks;
    for (;;) {
      if (generation_.load(std::memory_order_acquire) == seen) {
        if (ticks() - idle_since < spin_for) {__builtin_ia32_pause();} else {futex_wait(&generation_, seen);}
        continue;
      }
      if (stop_.load(std::memory_order_acquire)) {return;}
      // every worker checks in for every generation: run() does not return (and no new generation starts) before all have
      ++seen;
      work();
      spin_for = std::max<uint64_t>(kSpinTicks, linger_.load(std::memory_order_relaxed));       // (read before the check-in: run() may not have returned yet)
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {futex_wake_all(&pending_);}
      idle_since = ticks();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex run_mu_;
  const std::function<void(size_t)> * fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  std::atomic<uint32_t> pending_{0}, generation_{0};
  std::atomic<bool> stop_{false}, failed_{false};
  std::atomic<uint64_t> linger_{0};
  std::exception_ptr error_;
};

}  // namespace kh
