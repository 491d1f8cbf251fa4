// CPU probe of the solver's symbolic analysis (no GPU): reads a binary edge list (int32 pairs, node ids 0..N-1, node 0 = gauge)
// and prints the assembly tree level by level.  Build: g++ -O2 -std=c++17 -pthread tools/sym_probe.cpp slam_toolbox_amd/csrc/spa_symbolic.cpp -o /tmp/sym_probe
//   python -c "from slam_toolbox_amd import synth; import numpy as np; synth.make_pose_graph(10000,30000)['edges'].astype(np.int32).tofile('/tmp/edges.bin')"
//   /tmp/sym_probe /tmp/edges.bin 10000 [leaf] [pmax] [cands]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include "../slam_toolbox_amd/csrc/spa_symbolic.hpp"
#include "../slam_toolbox_amd/csrc/host_pool.hpp"
namespace kh {static std::string g_err; void set_error(const std::string & s) {g_err = s;}}
int main(int argc, char ** argv)
{
  if (argc < 3) {std::fprintf(stderr, "usage: sym_probe edges.bin n_nodes [leaf] [pmax] [cands]\n"); return 2;}
  FILE * f = std::fopen(argv[1], "rb");
  if (!f) {return 2;}
  std::vector<int32_t> e;
  int32_t buf[4096]; size_t n;
  while ((n = std::fread(buf, 4, 4096, f)) > 0) {e.insert(e.end(), buf, buf + n);}
  std::fclose(f);
  const int E = static_cast<int>(e.size() / 2), N = std::atoi(argv[2]);
  std::vector<std::vector<int>> adj(N - 1);
  for (int k = 0; k < E; ++k) {
    const int a = e[2 * k] - 1, b = e[2 * k + 1] - 1;
    if (a >= 0 && b >= 0 && a != b) {adj[a].push_back(b); adj[b].push_back(a);}
  }
  std::vector<int32_t> ptr(N, 0), idx;
  for (int i = 0; i < N - 1; ++i) {
    std::sort(adj[i].begin(), adj[i].end());
    adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
    idx.insert(idx.end(), adj[i].begin(), adj[i].end());
    ptr[i + 1] = static_cast<int32_t>(idx.size());
  }
  kh::SymbolicOptions opt;
  if (argc > 3) {opt.leaf_nodes = std::atoi(argv[3]);}
  if (argc > 4) {opt.max_pivot_nodes = std::atoi(argv[4]);}
  if (argc > 5) {opt.separator_candidates = std::atoi(argv[5]);}
  if (argc > 6) {opt.balance_lo = std::atof(argv[6]); opt.balance_hi = 1.0 - opt.balance_lo;}
  (void)argc;
  if (std::getenv("SYM_PROBE_THREADS")) {
    // the dissection's one parallel loop on plain threads (the library passes its persistent pool)
    const int nt = std::max(1, std::atoi(std::getenv("SYM_PROBE_THREADS")));
    opt.parallel_for = [nt](size_t n, const std::function<void(size_t)> & fn) {
      std::atomic<size_t> next{0};
      std::vector<std::thread> team;
      auto work = [&] {for (size_t i; (i = next.fetch_add(1)) < n;) {fn(i);}};
      for (int t = 1; t < nt; ++t) {team.emplace_back(work);}
      work();
      for (auto & t : team) {t.join();}
    };
  }
  if (std::getenv("SYM_PROBE_POOL")) {
    // ... or on the library's persistent pool (KH_HOST_THREADS), as kh_spa_compute passes it
    opt.parallel_for = [](size_t n, const std::function<void(size_t)> & fn) {kh::HostPool::instance().run(n, fn);};
  }
  kh::Symbolic sym;
  double best = 1e30;
  int rc = 0;
  const int reps = std::getenv("SYM_PROBE_REPS") ? std::atoi(std::getenv("SYM_PROBE_REPS")) : 5;
  for (int rep = 0; rep < reps; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    rc = kh::build_symbolic(sym, N - 1, ptr, idx, opt);
    best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  std::printf("rc %d (%s) fronts %d levels %zu nnz(L) %.2fM flops %.0fM front storage %.0f MB winv %.0f MB max m %d max ns %d  best: %.2f ms\n", rc,
    kh::g_err.c_str(), sym.n_fronts, sym.levels.size(), sym.nnz_factor / 1e6, sym.factor_flops / 1e6, sym.fronts_size * 8e-6, sym.winv_size * 8e-6,
    sym.max_m, sym.max_ns, best);
  int sum_ns = 0;
  for (size_t l = 0; l < sym.levels.size(); ++l) {
    int mm = 0, mns = 0; double work = 0;
    for (int k : sym.levels[l]) {
      mm = std::max(mm, sym.front_m[k]); mns = std::max(mns, sym.front_ns[k]);
      for (int j = 0; j < sym.front_ns[k]; ++j) {work += double(sym.front_m[k] - j) * (sym.front_m[k] - j);}
    }
    sum_ns += mns;
    std::printf("level %2zu: %4zu fronts, max m %3d, max ns %3d, %.1fM mult-adds\n", l, sym.levels[l].size(), mm, mns, work / 1e6);
  }
  std::printf("sum of the levels' largest pivot counts: %d\n", sum_ns);
  if (const char * dump = std::getenv("SYM_PROBE_DUMP")) {
    // int32 arrays, each preceded by its length: free_of_elim, front_first, front_ns, front_m, level, parent, rows_ptr, rows,
    // child_ptr, child_list, relpos_ptr, relpos, cinv_ptr, cinv
    FILE * o = std::fopen(dump, "wb");
    auto put = [&](const std::vector<int32_t> & v) {
      const int32_t n32 = static_cast<int32_t>(v.size());
      std::fwrite(&n32, 4, 1, o); std::fwrite(v.data(), 4, v.size(), o);
    };
    put(sym.free_of_elim); put(sym.front_first); put(sym.front_ns); put(sym.front_m); put(sym.level); put(sym.parent);
    put(sym.rows_ptr); put(sym.rows); put(sym.child_ptr); put(sym.child_list); put(sym.relpos_ptr); put(sym.relpos);
    put(sym.cinv_ptr); put(sym.cinv);
    std::fclose(o);
  }
  return rc;
}
