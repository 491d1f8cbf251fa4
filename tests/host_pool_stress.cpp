// Stress program for slam_toolbox_amd/csrc/host_pool.hpp (built and run by tests/test_host_pool.py, no GPU):
// parallel regions of random sizes from two caller threads, back to back (workers still spinning) and with pauses (workers
// asleep on the futex), every index of every region executed exactly once; nested regions; exceptions out of a region.
#include <chrono>
#include <cstdio>
#include <random>
#include <stdexcept>

#include "../slam_toolbox_amd/csrc/host_pool.hpp"

static int run_regions(unsigned seed, int regions, bool back_to_back, std::atomic<long long> & total)
{
  std::mt19937 rng(seed);
  kh::HostPool & pool = kh::HostPool::instance();
  int bad = 0;
  for (int r = 0; r < regions; ++r) {
    const size_t n = rng() % 97;
    std::vector<std::atomic<int>> hits(n);
    for (auto & h : hits) {h.store(0);}
    // (every seventh region asks the workers to stay awake behind it: the hint must change nothing but their sleep)
    pool.run(n, [&](size_t i) {hits[i].fetch_add(1); total.fetch_add(static_cast<long long>(i) + 1);}, (r % 7) == 0 ? 100000 : 0);
    for (size_t i = 0; i < n; ++i) {if (hits[i].load() != 1) {++bad;}}
    if (!back_to_back && (r % 16) == 0) {std::this_thread::sleep_for(std::chrono::microseconds(200));}      // let the workers fall asleep
  }
  return bad;
}

int main()
{
  std::atomic<long long> total{0};
  int bad = 0;
  for (int round = 0; round < 4; ++round) {
    const bool back_to_back = (round & 1) != 0;
    int bad_a = 0, bad_b = 0;
    // two callers: the one that finds the pool taken runs its region itself
    std::thread other([&] {bad_b = run_regions(100 + round, 2000, back_to_back, total);});
    bad_a = run_regions(200 + round, 2000, back_to_back, total);
    other.join();
    bad += bad_a + bad_b;
  }
  // the expected sum, recomputed serially
  long long expect = 0;
  for (int round = 0; round < 4; ++round) {
    for (unsigned seed : {100u + round, 200u + round}) {
      std::mt19937 rng(seed);
      for (int r = 0; r < 2000; ++r) {const long long n = rng() % 97; expect += n * (n + 1) / 2;}
    }
  }
  // a region entered from inside a region (on the caller thread and on workers) runs in line, every index once
  {
    kh::HostPool & pool = kh::HostPool::instance();
    std::atomic<long long> nested{0};
    for (int r = 0; r < 200; ++r) {
      pool.run(40, [&](size_t i) {pool.run(7, [&](size_t j) {nested.fetch_add(static_cast<long long>(i * 7 + j) + 1);});});
    }
    if (nested.load() != 200ll * (280ll * 281ll / 2)) {std::printf("nested regions: %lld\n", nested.load()); ++bad;}
  }
  // an exception thrown inside a region -- on whichever thread -- reaches the caller after the region has drained, and the pool
  // goes on working
  {
    kh::HostPool & pool = kh::HostPool::instance();
    int caught = 0;
    for (int r = 0; r < 100; ++r) {
      try {
        pool.run(64, [&](size_t i) {if (i == static_cast<size_t>(r % 64)) {throw std::runtime_error("boom");}});
      } catch (const std::runtime_error &) {
        ++caught;
      }
      std::atomic<int> ok{0};
      pool.run(33, [&](size_t) {ok.fetch_add(1);});
      if (ok.load() != 33) {++bad;}
    }
    if (caught != 100) {std::printf("exceptions caught: %d of 100\n", caught); ++bad;}
  }
  std::printf("bad %d total %lld expect %lld\n", bad, total.load(), expect);
  return (bad == 0 && total.load() == expect) ? 0 : 1;
}
