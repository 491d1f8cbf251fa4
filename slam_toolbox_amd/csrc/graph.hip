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
#include <cstdint>
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
  GraphDev g, const int32_t * __restrict__ queries, double max_sq_plus, double max_sq_minus, int32_t min_chain,
  uint8_t * flags_all, int32_t * frontier_all, int32_t * chain_count, int32_t * chains, int32_t cap_per_query)
{
  const int qi = blockIdx.x;
  const int q = queries[qi];
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
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const uint8_t f = flags[i];
    const bool good = (f & kInRange) && !(f & kLinked);
    if (!good) {continue;}
    bool emit;
    int len_needed;
    if (i == n - 1) {
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
    while (s > 0) {
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
  int rc = ensure(g->d_xy, g->cap_xy, std::max<size_t>(2 * n, 2)); if (rc) {return rc;}
  rc = ensure(g->d_adj_ptr, g->cap_ptr, n + 1); if (rc) {return rc;}
  rc = ensure(g->d_adj_idx, g->cap_idx, std::max<size_t>(n_adj, 1)); if (rc) {return rc;}
  if (n) {
    if (hipMemcpy(g->d_xy, ref_xy, 2 * n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(g->d_adj_ptr, adj_ptr, (n + 1) * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
      (n_adj && hipMemcpy(g->d_adj_idx, adj_idx, n_adj * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess))
    {
      set_error("kh_graph_set: upload failed");
      return KH_ERR_HIP;
    }
  }
  g->n = n_scans;
  return KH_OK;
}

int kh_graph_set_positions(kh_graph * g, int32_t n_scans, const double * ref_xy)
{
  if (!g || !ref_xy || n_scans != g->n) {return KH_ERR_INVALID_ARG;}
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  if (n_scans && hipMemcpy(g->d_xy, ref_xy, 2 * static_cast<size_t>(n_scans) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    set_error("kh_graph_set_positions: upload failed");
    return KH_ERR_HIP;
  }
  return KH_OK;
}

int kh_graph_find_loop_candidates(
  kh_graph * g, int32_t n_queries, const int32_t * query_scans, double max_distance, int32_t min_chain_size,
  int32_t * chain_begin, int32_t * chains, int32_t cap_chains, int32_t * n_chains)
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
  if (hipSetDevice(g->device) != hipSuccess) {return KH_ERR_HIP;}
  const size_t nq = static_cast<size_t>(n_queries), n = static_cast<size_t>(g->n);
  // a run needs at least one terminator, so a query has at most n / 2 + 1 chains; min_chain bounds it further
  const int32_t per_query = static_cast<int32_t>(std::min<size_t>(n / std::max(1, min_chain_size + 1) + 2, n / 2 + 1));
  int rc = ensure(g->d_queries, g->cap_q, nq); if (rc) {return rc;}
  rc = ensure(g->d_flags, g->cap_flags, nq * ((n + 3) & ~static_cast<size_t>(3)) + 4); if (rc) {return rc;}
  rc = ensure(g->d_frontier, g->cap_frontier, nq * 2 * n); if (rc) {return rc;}
  rc = ensure(g->d_count, g->cap_count, nq); if (rc) {return rc;}
  rc = ensure(g->d_chains, g->cap_chains, nq * per_query * 2); if (rc) {return rc;}
  if (hipMemcpyAsync(g->d_queries, query_scans, nq * sizeof(int32_t), hipMemcpyHostToDevice, g->stream) != hipSuccess) {return KH_ERR_HIP;}
  // Mapper.cpp:1988-1990 and 1326-1327: Square(maxDistance) +/- KT_TOLERANCE
  const double sq = max_distance * max_distance;
  GraphDev dev{g->n, g->d_xy, g->d_adj_ptr, g->d_adj_idx};
  (void)hipEventRecord(g->ev[0], g->stream);
  hipLaunchKernelGGL(k_loop_candidates, dim3(static_cast<unsigned>(nq)), dim3(256), 0, g->stream, dev, g->d_queries,
    sq + kTol, sq - kTol, min_chain_size, g->d_flags, g->d_frontier, g->d_count, g->d_chains, per_query);
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

}  // extern "C"
