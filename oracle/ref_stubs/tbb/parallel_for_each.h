// Stand-in for tbb::parallel_for_each: a std::thread fork-join over the container,
// row-parallel like the original call site. Thread count from KARTO_REF_THREADS (default 1).
#pragma once
#include <atomic>
#include <cstdlib>
#include <thread>
#include <vector>
namespace tbb {
inline int & ref_thread_count() { static int n = 1; return n; }
template <class C, class F>
void parallel_for_each(C & c, const F & f)
{
  const int nt = ref_thread_count();
  if (nt <= 1) { for (auto it = c.begin(); it != c.end(); ++it) { f(*it); } return; }
  std::atomic<size_t> next(0);
  const size_t n = c.size();
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) {
    pool.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < n; ) { f(c[i]); } });
  }
  for (auto & th : pool) { th.join(); }
}
}
