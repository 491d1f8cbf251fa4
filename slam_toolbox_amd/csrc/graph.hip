// Loop-closure candidate enumeration on the GPU: the part of
// karto::MapperGraph that feeds the batched scan matcher.
//
//   FindNearLinkedScans      Mapper.cpp:1795-1806 (BreadthFirstTraversal Mapper.cpp:1263-1297 with
//                            NearScanVisitor Mapper.cpp:1311-1333 over Vertex::GetAdjacentVertices Mapper.h:338-361)
//   FindPossibleLoopClosure  Mapper.cpp:1960-2010, all the chains successive calls return (TryCloseLoop's
//                            enumeration loop, Mapper.cpp:1500-1560), for a BATCH of query scans at once
//
// The reference walks all N scans per query on the CPU (distance test per scan) and runs a BFS with
// std::set / std::find membership tests; with matching at ~30 us per candidate that walk becomes the
// bottleneck.  Here the graph store (reference positions + CSR adjacency) is resident in HBM and one
// workgroup per query does: (1) the two distance tests for every scan -- the same IEEE operations as the
// reference (dx*dx + dy*dy, compiled without FMA contraction), so the flags are bit-exact; (2) the BFS
// restricted to "visitable" vertices, level-synchronous, frontier in global scratch; (3) the run
// segmentation that replaces the sequential chain state machine: a maximal run of (in range, not linked)
// scans is a chain iff it is terminated by an out-of-range scan and is long enough, or by the end of the
// list; a run terminated by a linked scan is discarded (chain.clear(), Mapper.cpp:1993-1996).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/karto_hip.h"

namespace kh
{
void set_error(const std::string & s);

constexpr double kTol = 1e-06;      // KT_TOLERANCE, Math.h:41
constexpr uint8_t kInRange = 1;     // squaredDistance <  maxDistance^2 + KT_TOLERANCE   (Mapper.cpp:1988-1990)
constexpr uint8_t kVisitable = 2;   // squaredDistance <= maxDistance^2 - KT_TOLERANCE   (Mapper.cpp:1326-1327)
constexpr uint8_t kLinked = 4;      // member of FindNearLinkedScans' result
constexpr uint8_t kSeen = 8;

struct GraphDev
{
  int32_t n;
  const double * xy;
  const int32_t * adj_ptr;
  const int32_t * adj_idx;
};

__global__ __launch_bounds__(256) void k_loop_candidates(
  GraphDev g, const int32_t * __restrict__ queries, const int32_t * __restrict__ starts, double max_sq_plus, double max_sq_minus,
  int32_t min_chain, int32_t n_visit, uint8_t * flags_all, int32_t * frontier_all, int32_t * chain_count, int32_t * chains, int32_t cap_per_query)
{
  const int qi = blockIdx.x;
  const int q = queries[qi];
  // FindPossibleLoopClosure resumes at rStartNum with an EMPTY chain (Mapper.cpp:1966, 1976): scans before it neither
  // form chains nor extend one
  const int start = starts ? starts[qi] : 0;
  const int n = g.n;
  uint8_t * flags = flags_all + (size_t)qi * ((n + 3) & ~3);       // word-aligned rows: bits are set with 32-bit atomics
  int32_t * cur = frontier_all + (size_t)qi * 2 * n;
  int32_t * nxt = cur + n;
  __shared__ int32_t s_cur_n, s_nxt_n, s_out_n;
  const double qx = g.xy[2 * q], qy = g.xy[2 * q + 1];
  // (1) distance flags
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double dx = g.xy[2 * i] - qx, dy = g.xy[2 * i + 1] - qy;
    const double d2 = dx * dx + dy * dy;
    uint8_t f = 0;
    if (d2 < max_sq_plus) {f |= kInRange;}
    if (d2 <= max_sq_minus) {f |= kVisitable;}
    flags[i] = f;
  }
  if (threadIdx.x == 0) {s_cur_n = 1; s_nxt_n = 0; s_out_n = 0; cur[0] = q;}
  __syncthreads();
  if (threadIdx.x == 0) {flags[q] |= kSeen;}
  __syncthreads();
  // (2) BFS over visitable vertices: a popped vertex is valid iff visitable, only valid ones expand
  while (s_cur_n > 0) {
    const int cn = s_cur_n;
    for (int t = threadIdx.x; t < cn; t += blockDim.x) {
      const int v = cur[t];
      if (!(flags[v] & kVisitable)) {continue;}
      atomicOr(reinterpret_cast<unsigned int *>(flags + (v & ~3)), (unsigned int)kLinked << (8 * (v & 3)));
      for (int k = g.adj_ptr[v]; k < g.adj_ptr[v + 1]; ++k) {
        const int w = g.adj_idx[k];
        const unsigned int bit = (unsigned int)kSeen << (8 * (w & 3));
        const unsigned int old = atomicOr(reinterpret_cast<unsigned int *>(flags + (w & ~3)), bit);
        if (!(old & bit)) {nxt[atomicAdd(&s_nxt_n, 1)] = w;}
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {s_cur_n = s_nxt_n; s_nxt_n = 0;}
    int32_t * tmp = cur; cur = nxt; nxt = tmp;
    __syncthreads();
  }
  // (3) chains = maximal runs of good scans with the right terminator
  int32_t * out = chains + (size_t)qi * cap_per_query * 2;
  // the walk ends at n_visit (the reference's loop bound is the scan MAP's size, which falls behind the largest id
  // once scans have been removed, Mapper.cpp:1974-1976): whatever chain is open there is returned
  for (int i = threadIdx.x; i < n_visit; i += blockDim.x) {
    const uint8_t f = flags[i];
    const bool good = i >= start && (f & kInRange) && !(f & kLinked);
    if (!good) {continue;}
    bool emit;
    int len_needed;
    if (i == n_visit - 1) {
      emit = true; len_needed = 1;                       // end of the list: whatever is left is returned
    } else {
      const uint8_t fn = flags[i + 1];
      const bool next_good = (fn & kInRange) && !(fn & kLinked);
      if (next_good) {continue;}                         // not the end of its run
      emit = !(fn & kInRange);                           // out of range: chain returned if long enough; linked: cleared
      len_needed = min_chain;
    }
    if (!emit) {continue;}
    int s = i;
    while (s > start) {
      const uint8_t fp = flags[s - 1];
      if ((fp & kInRange) && !(fp & kLinked)) {--s;} else {break;}
    }
    if (i - s + 1 >= len_needed) {
      const int slot = atomicAdd(&s_out_n, 1);
      if (slot < cap_per_query) {out[2 * slot] = s; out[2 * slot + 1] = i;}
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {chain_count[qi] = s_out_n;}
}

template <class T>
static int ensure(T *& p, size_t & cap, size_t need)
{
  if (need <= cap) {return KH_OK;}
  if (p) {(void)hipFree(p); p = nullptr;}
  const size_t n = std::max(need, cap + cap / 2);
  if (hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)) != hipSuccess) {set_error("hipMalloc failed (graph store)"); cap = 0; return KH_ERR_HIP;}
  cap = n;
  return KH_OK;
}

}  // namespace kh

using namespace kh;

struct kh_graph
{
  int32_t device = 0;
  hipStream_t stream = nullptr;
  int32_t n = 0;
  bool device_stale = false;   // the host copy is newer than the device arrays (uploaded by the next enumeration kernel)
  int32_t n_visit = 0;     // scans the candidate walks visit (kh_graph_set_scan_limit; = n unless scans were removed)
  double * d_xy = nullptr; size_t cap_xy = 0;
  int32_t * d_adj_ptr = nullptr; size_t cap_ptr = 0;
  int32_t * d_adj_idx = nullptr; size_t cap_idx = 0;
  int32_t * d_queries = nullptr; size_t cap_q = 0;
  uint8_t * d_flags = nullptr; size_t cap_flags = 0;
  int32_t * d_frontier = nullptr; size_t cap_frontier = 0;
  int32_t * d_count = nullptr; size_t cap_count = 0;
  int32_t * d_chains = nullptr; size_t cap_chains = 0;
  double last_ms = 0.0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  // host copy of the store: the neighbourhood walks of FindNearChains touch tens of vertices (no kernel)
  std::vector<double> h_xy;
  std::vector<int32_t> h_adj_ptr, h_adj_idx;
};

extern "C" {

int kh_graph_create(int32_t device, kh_graph ** out)
{
  if (!out) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  kh_graph * g = new kh_graph();
  g->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
    hipEventCreate(&g->ev[0]) != hipSuccess || hipEventCreate(&g->ev[1]) != hipSuccess)
  {
    set_error("HIP stream/event creation failed");
    delete g;
    return KH_ERR_HIP;
  }
  *out = g;
  return KH_OK;
}

void kh_graph_destroy(kh_graph * g)
{
  if (!g) {return;}
  (void)hipSetDevice(g->device);
  if (g->stream) {(void)hipStreamSynchronize(g->stream);}
  (void)hipFree(g->d_xy); (void)hipFree(g->d_adj_ptr); (void)hipFree(g->d_adj_idx); (void)hipFree(g->d_queries);
  (void)hipFree(g->d_flags); (void)hipFree(g->d_frontier); (void)hipFree(g->d_count); (void)hipFree(g->d_chains);
  if (g->ev[0]) {(void)hipEventDestroy(g->ev[0]);}
  if (g->ev[1]) {(void)hipEventDestroy(g->ev[1]);}
  if (g->stream) {(void)hipStreamDestroy(g->stream);}
  delete g;
}

int kh_graph_set(kh_graph * g, int32_t n_scans, const double * ref_xy, const int32_t * adj_ptr, const int32_t * adj_idx)
{
  if (!g || n_scans < 0 || (n_scans > 0 && (!ref_xy || !adj_ptr))) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  const size_t n = static_cast<size_t>(n_scans);
  const size_t n_adj = n ? static_cast<size_t>(adj_ptr[n]) : 0;
  if (n_adj > 0 && !adj_idx) {return KH_ERR_INVALID_ARG;}
  for (size_t k = 0; k < n_adj; ++k) {
    if (adj_idx[k] < 0 || adj_idx[k] >= n_scans) {set_error("kh_graph_set: adjacency index out of range"); return KH_ERR_INVALID_ARG;}
  }
  // the neighbourhood walks (near chains, near linked) read the host copy; the device arrays are refreshed when an
  // enumeration kernel next needs them (a mapper sets the graph several times per scan and enumerates once)
  g->device_stale = true;
  g->n = n_scans; g->n_visit = n_scans;
  g->h_xy.assign(ref_xy, ref_xy + 2 * n);
  g->h_adj_ptr.assign(adj_ptr, adj_ptr + (n ? n + 1 : 0));
  g->h_adj_idx.assign(adj_idx, adj_idx + n_adj);
  return KH_OK;
}

}  // extern "C"
namespace kh
{
// (library-internal, the mapper's sync_graph) kh_graph_set without the copies: the store takes the caller's arrays and hands its old
// ones back (same capacity next time); the caller built adj_idx from its own tables, so the range check is skipped.  A lifelong mapper
// rebuilds the store after every node removal -- once per accepted scan, 18 000 scans alive in the 50 000-scan replay.
int graph_swap(kh_graph * g, int32_t n_scans, std::vector<double> & ref_xy, std::vector<int32_t> & adj_ptr, std::vector<int32_t> & adj_idx)
{
  if (!g || n_scans < 0 || ref_xy.size() != 2 * static_cast<size_t>(n_scans) || adj_ptr.size() != static_cast<size_t>(n_scans) + 1) {return KH_ERR_INVALID_ARG;}
  g->device_stale = true;
  g->n = n_scans; g->n_visit = n_scans;
  g->h_xy.swap(ref_xy); g->h_adj_ptr.swap(adj_ptr); g->h_adj_idx.swap(adj_idx);
  return KH_OK;
}
}  // namespace kh
extern "C" {

int kh_graph_append_scan(kh_graph * g, const double ref_xy[2])
{
  if (!g || !ref_xy) {return KH_ERR_INVALID_ARG;}
  if (g->h_adj_ptr.empty()) {g->h_adj_ptr.push_back(0);}
  g->h_xy.push_back(ref_xy[0]); g->h_xy.push_back(ref_xy[1]);
  g->h_adj_ptr.push_back(g->h_adj_ptr.back());
  if (g->n_visit == g->n) {++g->n_visit;}
  ++g->n;
  g->device_stale = true;
  return KH_OK;
}

int kh_graph_add_edge(kh_graph * g, int32_t scan_a, int32_t scan_b)
{
  if (!g || scan_a < 0 || scan_b < 0 || scan_a >= g->n || scan_b >= g->n || scan_a == scan_b) {return KH_ERR_INVALID_ARG;}
  auto append = [&](int32_t at, int32_t what) {
    g->h_adj_idx.insert(g->h_adj_idx.begin() + g->h_adj_ptr[static_cast<size_t>(at) + 1], what);
    for (size_t k = static_cast<size_t>(at) + 1; k < g->h_adj_ptr.size(); ++k) {++g->h_adj_ptr[k];}
  };
  append(scan_a, scan_b);
  append(scan_b, scan_a);
  g->device_stale = true;
  return KH_OK;
}

int kh_graph_set_position(kh_graph * g, int32_t scan, const double ref_xy[2])
{
  if (!g || !ref_xy || scan < 0 || scan >= g->n) {return KH_ERR_INVALID_ARG;}
  g->h_xy[2 * static_cast<size_t>(scan)] = ref_xy[0]; g->h_xy[2 * static_cast<size_t>(scan) + 1] = ref_xy[1];
  g->device_stale = true;
  return KH_OK;
}

int kh_graph_set_positions(kh_graph * g, int32_t n_scans, const double * ref_xy)
{
  if (!g || !ref_xy || n_scans != g->n) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  g->h_xy.assign(ref_xy, ref_xy + 2 * static_cast<size_t>(n_scans));
  g->device_stale = true;
  return KH_OK;
}

int kh_graph_find_loop_candidates(
  kh_graph * g, int32_t n_queries, const int32_t * query_scans, double max_distance, int32_t min_chain_size,
  int32_t * chain_begin, int32_t * chains, int32_t cap_chains, int32_t * n_chains)
{
  return kh_graph_find_loop_candidates_from(g, n_queries, query_scans, nullptr, max_distance, min_chain_size, chain_begin, chains,
           cap_chains, n_chains);
}

// One query, answered from the host copy of the store: the same three steps as k_loop_candidates -- the same IEEE
// operations for the two distance tests, the breadth-first marking of the linked scans, the run rule per scan -- in ~20 us
// for an 18 000-scan store, where the device round trip (upload of the edited store, launch, two downloads, a stream drain)
// is ~150 us.  A mapper asks exactly one such question per processed scan; batches of queries go to the kernel.
static void loop_candidates_host(const kh_graph * g, int32_t q, int32_t start, double max_sq_plus, double max_sq_minus, int32_t min_chain,
                                 std::vector<std::pair<int32_t, int32_t>> & out)
{
  const int32_t n = g->n, n_visit = g->n_visit;
  static thread_local std::vector<uint8_t> flags;
  static thread_local std::vector<int32_t> cur, nxt;
  flags.assign(static_cast<size_t>(n), 0);
  const double * xy = g->h_xy.data();
  const double qx = xy[2 * q], qy = xy[2 * q + 1];
  for (int32_t i = 0; i < n; ++i) {
    const double dx = xy[2 * i] - qx, dy = xy[2 * i + 1] - qy;
    const double d2 = dx * dx + dy * dy;
    uint8_t f = 0;
    if (d2 < max_sq_plus) {f |= kInRange;}
    if (d2 <= max_sq_minus) {f |= kVisitable;}
    flags[i] = f;
  }
  cur.assign(1, q); nxt.clear();
  flags[q] |= kSeen;
  while (!cur.empty()) {
    for (int32_t v : cur) {
      if (!(flags[v] & kVisitable)) {continue;}
      flags[v] |= kLinked;
      for (int32_t k = g->h_adj_ptr[v]; k < g->h_adj_ptr[v + 1]; ++k) {
        const int32_t w = g->h_adj_idx[k];
        if (!(flags[w] & kSeen)) {flags[w] |= kSeen; nxt.push_back(w);}
      }
    }
    cur.swap(nxt);
    nxt.clear();
  }
  out.clear();
  auto good_at = [&](int32_t i) {return (flags[i] & kInRange) && !(flags[i] & kLinked);};
  for (int32_t i = std::max(start, 0); i < n_visit; ++i) {
    if (!good_at(i)) {continue;}
    bool emit;
    int32_t len_needed;
    if (i == n_visit - 1) {
      emit = true; len_needed = 1;                       // end of the list: whatever is left is returned
    } else {
      if (good_at(i + 1)) {continue;}                    // not the end of its run
      emit = !(flags[i + 1] & kInRange);                 // out of range: chain returned if long enough; linked: cleared
      len_needed = min_chain;
    }
    if (!emit) {continue;}
    int32_t s0 = i;
    while (s0 > start && good_at(s0 - 1)) {--s0;}
    if (i - s0 + 1 >= len_needed) {out.push_back({s0, i});}
  }
}

int kh_graph_find_loop_candidates_from(
  kh_graph * g, int32_t n_queries, const int32_t * query_scans, const int32_t * start_scans, double max_distance,
  int32_t min_chain_size, int32_t * chain_begin, int32_t * chains, int32_t cap_chains, int32_t * n_chains)
{
  if (!g || n_queries < 0 || !chain_begin || !n_chains || (n_queries > 0 && !query_scans) || cap_chains < 0 || (cap_chains > 0 && !chains)) {
    return KH_ERR_INVALID_ARG;
  }
  *n_chains = 0;
  chain_begin[0] = 0;
  if (n_queries == 0) {return KH_OK;}
  if (g->n <= 0) {set_error("kh_graph_find_loop_candidates: empty graph"); return KH_ERR_NOT_FOUND;}
  for (int32_t i = 0; i < n_queries; ++i) {
    if (query_scans[i] < 0 || query_scans[i] >= g->n) {set_error("kh_graph_find_loop_candidates: unknown scan"); return KH_ERR_NOT_FOUND;}
  }
  static const bool host_single = !(std::getenv("KH_GRAPH_HOST_SINGLE") && std::atoi(std::getenv("KH_GRAPH_HOST_SINGLE")) == 0);
  if (n_queries == 1 && host_single) {
    if (start_scans && start_scans[0] < 0) {set_error("kh_graph_find_loop_candidates_from: negative start"); return KH_ERR_INVALID_ARG;}
    const double sq1 = max_distance * max_distance;
    std::vector<std::pair<int32_t, int32_t>> v;
    loop_candidates_host(g, query_scans[0], start_scans ? start_scans[0] : 0, sq1 + kTol, sq1 - kTol, min_chain_size, v);
    int32_t total1 = 0;
    for (const auto & ch : v) {                            // already in scan order
      if (total1 < cap_chains) {chains[2 * total1] = ch.first; chains[2 * total1 + 1] = ch.second;}
      ++total1;
    }
    chain_begin[1] = total1;
    *n_chains = total1;
    g->last_ms = 0.0;
    return KH_OK;
  }
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  if (g->device_stale) {
    const size_t ns = static_cast<size_t>(g->n), n_adj = g->h_adj_idx.size();
    int urc = ensure(g->d_xy, g->cap_xy, std::max<size_t>(2 * ns, 2)); if (urc) {return urc;}
    urc = ensure(g->d_adj_ptr, g->cap_ptr, ns + 1); if (urc) {return urc;}
    urc = ensure(g->d_adj_idx, g->cap_idx, std::max<size_t>(n_adj, 1)); if (urc) {return urc;}
    if (hipMemcpyAsync(g->d_xy, g->h_xy.data(), 2 * ns * sizeof(double), hipMemcpyHostToDevice, g->stream) != hipSuccess ||
      hipMemcpyAsync(g->d_adj_ptr, g->h_adj_ptr.data(), (ns + 1) * sizeof(int32_t), hipMemcpyHostToDevice, g->stream) != hipSuccess ||
      (n_adj && hipMemcpyAsync(g->d_adj_idx, g->h_adj_idx.data(), n_adj * sizeof(int32_t), hipMemcpyHostToDevice, g->stream) != hipSuccess) ||
      hipStreamSynchronize(g->stream) != hipSuccess)
    {
      set_error("kh_graph: upload failed");
      return KH_ERR_HIP;
    }
    g->device_stale = false;
  }
  const size_t nq = static_cast<size_t>(n_queries), n = static_cast<size_t>(g->n);
  // a run needs at least one terminator, so a query has at most n / 2 + 1 chains; min_chain bounds it further
  const int32_t per_query = static_cast<int32_t>(std::min<size_t>(n / std::max(1, min_chain_size + 1) + 2, n / 2 + 1));
  int rc = ensure(g->d_queries, g->cap_q, 2 * nq); if (rc) {return rc;}
  rc = ensure(g->d_flags, g->cap_flags, nq * ((n + 3) & ~static_cast<size_t>(3)) + 4); if (rc) {return rc;}
  rc = ensure(g->d_frontier, g->cap_frontier, nq * 2 * n); if (rc) {return rc;}
  rc = ensure(g->d_count, g->cap_count, nq); if (rc) {return rc;}
  rc = ensure(g->d_chains, g->cap_chains, nq * per_query * 2); if (rc) {return rc;}
  if (hipMemcpyAsync(g->d_queries, query_scans, nq * sizeof(int32_t), hipMemcpyHostToDevice, g->stream) != hipSuccess) {return KH_ERR_HIP;}
  if (start_scans) {
    for (size_t i = 0; i < nq; ++i) {if (start_scans[i] < 0) {set_error("kh_graph_find_loop_candidates_from: negative start"); return KH_ERR_INVALID_ARG;}}
    if (hipMemcpyAsync(g->d_queries + nq, start_scans, nq * sizeof(int32_t), hipMemcpyHostToDevice, g->stream) != hipSuccess) {return KH_ERR_HIP;}
  }
  // Mapper.cpp:1988-1990 and 1326-1327: Square(maxDistance) +/- KT_TOLERANCE
  const double sq = max_distance * max_distance;
  GraphDev dev{g->n, g->d_xy, g->d_adj_ptr, g->d_adj_idx};
  (void)hipEventRecord(g->ev[0], g->stream);
  hipLaunchKernelGGL(k_loop_candidates, dim3(static_cast<unsigned>(nq)), dim3(256), 0, g->stream, dev, g->d_queries,
    start_scans ? g->d_queries + nq : nullptr, sq + kTol, sq - kTol, min_chain_size, g->n_visit, g->d_flags, g->d_frontier, g->d_count, g->d_chains, per_query);
  (void)hipEventRecord(g->ev[1], g->stream);
  std::vector<int32_t> counts(nq), all(nq * per_query * 2);
  if (hipMemcpyAsync(counts.data(), g->d_count, nq * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
    hipMemcpyAsync(all.data(), g->d_chains, all.size() * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
    hipStreamSynchronize(g->stream) != hipSuccess)
  {
    set_error(std::string("kh_graph_find_loop_candidates: ") + hipGetErrorString(hipGetLastError()));
    return KH_ERR_HIP;
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, g->ev[0], g->ev[1]);
  g->last_ms = ms;
  int32_t total = 0;
  for (size_t qi = 0; qi < nq; ++qi) {
    const int32_t c = std::min(counts[qi], per_query);
    std::vector<std::pair<int32_t, int32_t>> v(c);
    for (int32_t k = 0; k < c; ++k) {v[k] = {all[(qi * per_query + k) * 2], all[(qi * per_query + k) * 2 + 1]};}
    std::sort(v.begin(), v.end());                       // the reference returns them in scan order
    for (const auto & ch : v) {
      if (total < cap_chains) {chains[2 * total] = ch.first; chains[2 * total + 1] = ch.second;}
      ++total;
    }
    chain_begin[qi + 1] = total;
  }
  *n_chains = total;
  return KH_OK;
}

double kh_graph_last_kernel_ms(kh_graph * g) {return g ? g->last_ms : 0.0;}

int kh_graph_set_scan_limit(kh_graph * g, int32_t n_visit)
{
  if (!g || n_visit < 0 || n_visit > g->n) {return KH_ERR_INVALID_ARG;}
  g->n_visit = n_visit;
  return KH_OK;
}

// ---- host-side members of the row (exact reference arithmetic, O(neighbourhood) work) ---------------------
namespace
{
constexpr double kTolerance = 1e-06;      // KT_TOLERANCE, Math.h:41

inline double squared_distance(const double * a, const double * b)     // Vector2::SquaredDistance
{
  const double dx = a[0] - b[0], dy = a[1] - b[1];
  return dx * dx + dy * dy;
}

// BreadthFirstTraversal::TraverseForVertices with NearScanVisitor (Mapper.cpp:1263-1297, 1311-1333): valid
// vertices in visit order; the start vertex is visited like any other
void near_linked(const kh_graph * g, int32_t q, double max_distance, std::vector<int32_t> & valid)
{
  const double lim = max_distance * max_distance - kTolerance;       // Visit(): squaredDistance <= max^2 - tol
  const double * centre = &g->h_xy[2 * static_cast<size_t>(q)];
  std::vector<int32_t> queue(1, q);
  std::vector<uint8_t> seen(static_cast<size_t>(g->n), 0);
  seen[q] = 1;
  for (size_t h = 0; h < queue.size(); ++h) {
    const int32_t v = queue[h];
    if (squared_distance(&g->h_xy[2 * static_cast<size_t>(v)], centre) <= lim) {
      valid.push_back(v);
      for (int32_t k = g->h_adj_ptr[v]; k < g->h_adj_ptr[v + 1]; ++k) {
        const int32_t w = g->h_adj_idx[k];
        if (!seen[w]) {seen[w] = 1; queue.push_back(w);}
      }
    }
  }
}

void matrix3_inverse(const double * m, double * inv)    // Karto.h:2533-2577 (row-major 3x3)
{
  inv[0] = m[4] * m[8] - m[5] * m[7];
  inv[1] = m[2] * m[7] - m[1] * m[8];
  inv[2] = m[1] * m[5] - m[2] * m[4];
  inv[3] = m[5] * m[6] - m[3] * m[8];
  inv[4] = m[0] * m[8] - m[2] * m[6];
  inv[5] = m[2] * m[3] - m[0] * m[5];
  inv[6] = m[3] * m[7] - m[4] * m[6];
  inv[7] = m[1] * m[6] - m[0] * m[7];
  inv[8] = m[0] * m[4] - m[1] * m[3];
  const double det = m[0] * inv[0] + m[1] * inv[3] + m[2] * inv[6];
  if (std::fabs(det) <= 1e-14) {return;}      // assert(false) is compiled out in Release
  const double inv_det = 1.0 / det;
  for (int i = 0; i < 9; ++i) {inv[i] *= inv_det;}
}

double normalize_angle(double angle)   // math::NormalizeAngle, Math.h:181-202
{
  const double pi = 3.14159265358979323846, two_pi = 6.28318530717958647692;
  while (angle < -pi) {
    if (angle < -two_pi) {angle += static_cast<uint32_t>(angle / -two_pi) * two_pi;} else {angle += two_pi;}
  }
  while (angle > pi) {
    if (angle > two_pi) {angle -= static_cast<uint32_t>(angle / two_pi) * two_pi;} else {angle -= two_pi;}
  }
  return angle;
}
}  // namespace

int kh_graph_find_near_chains(
  kh_graph * g, int32_t query_scan, double link_scan_maximum_distance, int32_t * chains, int32_t cap_chains,
  int32_t * n_chains)
{
  if (!g || !n_chains || query_scan < 0 || query_scan >= g->n || (cap_chains > 0 && !chains)) {return KH_ERR_INVALID_ARG;}
  const double * pose = &g->h_xy[2 * static_cast<size_t>(query_scan)];
  const double lim = link_scan_maximum_distance * link_scan_maximum_distance + kTolerance;     // Mapper.cpp:1735-1737
  std::vector<int32_t> linked;
  near_linked(g, query_scan, link_scan_maximum_distance, linked);
  std::vector<uint8_t> processed(static_cast<size_t>(g->n), 0);
  int32_t total = 0;
  for (int32_t near : linked) {
    if (near == query_scan || processed[near]) {continue;}
    processed[near] = 1;
    bool valid = true;
    int32_t first = near, last = near;
    for (int32_t c = near - 1; c >= 0; --c) {                        // scans before (Mapper.cpp:1715-1746)
      if (c == query_scan) {valid = false;}
      if (squared_distance(pose, &g->h_xy[2 * static_cast<size_t>(c)]) < lim) {first = c; processed[c] = 1;} else {break;}
    }
    for (int32_t c = near + 1; c < g->n_visit; ++c) {                // scans after (Mapper.cpp:1751-1780); bound = scan map size
      if (c == query_scan) {valid = false;}
      if (squared_distance(pose, &g->h_xy[2 * static_cast<size_t>(c)]) < lim) {last = c; processed[c] = 1;} else {break;}
    }
    if (valid) {
      if (total < cap_chains) {chains[2 * total] = first; chains[2 * total + 1] = last;}
      ++total;
    }
  }
  *n_chains = total;
  return KH_OK;
}

int kh_graph_find_near_linked(kh_graph * g, int32_t query_scan, double max_distance, int32_t * scans, int32_t cap, int32_t * n_found)
{
  if (!g || !n_found || query_scan < 0 || query_scan >= g->n || (cap > 0 && !scans)) {return KH_ERR_INVALID_ARG;}
  std::vector<int32_t> valid;
  near_linked(g, query_scan, max_distance, valid);
  *n_found = static_cast<int32_t>(valid.size());
  for (int32_t k = 0; k < *n_found && k < cap; ++k) {scans[k] = valid[k];}
  return KH_OK;
}

int kh_graph_closest_scan_to_pose(kh_graph * g, const int32_t * scans, int32_t n, const double pose_xy[2], int32_t * closest)
{
  if (!g || !closest || !pose_xy || n < 0 || (n > 0 && !scans)) {return KH_ERR_INVALID_ARG;}
  int32_t best = -1;
  double best_d = 1.7976931348623157e308;                            // DBL_MAX
  for (int32_t k = 0; k < n; ++k) {
    if (scans[k] < 0 || scans[k] >= g->n) {return KH_ERR_INVALID_ARG;}
    const double d = squared_distance(pose_xy, &g->h_xy[2 * static_cast<size_t>(scans[k])]);
    if (d < best_d) {best_d = d; best = scans[k];}
  }
  *closest = best;                                                   // NULL (-1) for an empty chain
  return KH_OK;
}

int kh_weighted_mean(int32_t n, const double * means, const double * covariances, double mean[3])
{
  if (n <= 0 || !means || !covariances || !mean) {return KH_ERR_INVALID_ARG;}
  std::vector<double> inverses(9 * static_cast<size_t>(n));
  double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int32_t k = 0; k < n; ++k) {
    double * inv = &inverses[9 * static_cast<size_t>(k)];
    for (int i = 0; i < 9; ++i) {inv[i] = 0.0;}
    matrix3_inverse(covariances + 9 * static_cast<size_t>(k), inv);
    for (int i = 0; i < 9; ++i) {sum[i] += inv[i];}
  }
  double inv_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  matrix3_inverse(sum, inv_sum);
  double ax = 0.0, ay = 0.0, ah = 0.0, theta_x = 0.0, theta_y = 0.0;
  for (int32_t k = 0; k < n; ++k) {
    const double * p = means + 3 * static_cast<size_t>(k);
    const double * inv = &inverses[9 * static_cast<size_t>(k)];
    double sin_h, cos_h;
    ::sincos(p[2], &sin_h, &cos_h);          // one sincos like the reference's GCC build (see ref_sincos in matcher_host.cpp)
    theta_x += cos_h;
    theta_y += sin_h;
    double w[9];                                                     // inverseOfSumOfInverses * inverse, Karto.h:2634-2647
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        w[3 * r + c] = inv_sum[3 * r] * inv[c] + inv_sum[3 * r + 1] * inv[3 + c] + inv_sum[3 * r + 2] * inv[6 + c];
      }
    }
    // weight * pose (Karto.h:2654-2666), then Pose2::operator+= (Karto.h:2196-2200)
    ax += w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
    ay += w[3] * p[0] + w[4] * p[1] + w[5] * p[2];
    ah = normalize_angle(ah + (w[6] * p[0] + w[7] * p[1] + w[8] * p[2]));
  }
  theta_x /= static_cast<double>(static_cast<size_t>(n));
  theta_y /= static_cast<double>(static_cast<size_t>(n));
  (void)ah;
  mean[0] = ax; mean[1] = ay; mean[2] = std::atan2(theta_y, theta_x);
  return KH_OK;
}

}  // extern "C"
