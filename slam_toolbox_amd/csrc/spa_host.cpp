// Host side of the pose-graph SPA solver (hot path B): plugin state, symbolic analysis, and the
// Levenberg-Marquardt control loop; all numeric work runs on the GPU (spa_kernels.hip).
//
// Reference: solvers/ceres_solver.cpp (state + options + gauge), solvers/ceres_utils.h (residual),
// lib/karto_sdk/include/karto_sdk/Mapper.h:954-1066 (karto::ScanSolver), :174-188 (LinkInfo::Update),
// Karto.h:2533-2577 (Matrix3::Inverse).  The minimiser restates Ceres 2.0's trust-region LM
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, trust_region_step_evaluator.cc); Ceres
// itself is a third-party dependency that is not in the reference tree ("parity unpinned").
//
// Linear algebra design (MI355X): the normal matrix is factorised by a multifrontal block Cholesky.
// Ordering = geometric nested dissection on the node positions (a pose graph is a geometric graph:
// edges only link poses a few metres apart), which directly yields the supernodes (leaf clusters and
// separators); each supernode's frontal matrix is a small dense problem handled by one workgroup, and
// the elimination tree is processed level by level.
#include <hip/hip_runtime.h>

#include <immintrin.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <limits>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/karto_hip.h"
#include "spa_internal.hpp"
#include "spa_symbolic.hpp"

namespace kh
{
void set_error(const std::string & s);
void host_parallel_for(size_t n, const std::function<void(size_t)> & fn);      // matcher_host.cpp: the persistent host pool

#define KS_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error(std::string(#call) + ": " + hipGetErrorString(e_));                          \
      return KH_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

// `dead`: removed, waiting for settle() (RemoveNode / RemoveConstraint are O(1) / O(degree); a lifelong mapper removes a
// node every other scan, and erasing from the middle of 60 000 constraints + re-keying the maps each time cost ~1 ms per
// removed node)
struct Node {int32_t id; double pose[3]; uint8_t dead = 0;};
struct Constraint {int32_t a, b; double z[3]; double u[9]; double omega[6]; uint8_t dead = 0;};   // omega: upper triangle of the information

template <class T>
struct DevBuf
{
  typedef T value_type;
  T * p = nullptr; size_t cap = 0;
  bool owned = true;                 // false: p is a slice of another allocation (alias)
  // the buffer becomes n elements of somebody else's allocation (the analysis' arrays: slices of ONE block, one copy)
  void alias(T * slice, size_t n)
  {
    if (owned && p) {(void)hipFree(p);}
    p = slice; cap = n; owned = false;
  }
  int ensure(size_t n)
  {
    if (!owned) {p = nullptr; cap = 0; owned = true;}
    if (n <= cap) {return KH_OK;}
    if (p) {KS_HIP(hipFree(p)); p = nullptr;}
    cap = std::max(n, cap + cap / 2);
    KS_HIP(hipMalloc(reinterpret_cast<void **>(&p), cap * sizeof(T)));
    // KH_SPA_POISON=1 (debugging aid): new buffers start as NaN / -1 patterns, so that a kernel that reads what nobody
    // wrote shows up as a failed solve instead of as a last-bit difference between two processes
    static const bool poison = std::getenv("KH_SPA_POISON") != nullptr;
    if (poison) {KS_HIP(hipMemset(p, 0xFF, cap * sizeof(T))); KS_HIP(hipDeviceSynchronize());}
    return KH_OK;
  }
  int upload(const std::vector<T> & v, hipStream_t s)
  {
    int rc = ensure(std::max<size_t>(v.size(), 1));
    if (rc) {return rc;}
    if (!v.empty()) {KS_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));}
    return KH_OK;
  }
  void release() {if (p && owned) {(void)hipFree(p);} p = nullptr; cap = 0; owned = true;}
};

}  // namespace kh

using namespace kh;

struct kh_spa
{
  int32_t device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;               // the part of a level's extend-add that k_potrf does not read
  hipEvent_t ev_level[2] = {};
  // self-cleaning fronts (scatter mode): the buffers these pointers name are all zeros (every kernel of the last factorisation
  // and its backward sweep ran, each zeroing what it was the last to read)
  const double * clean_a = nullptr; const double * clean_b = nullptr;
  DevBuf<int32_t> d_deferred;                  // fronts whose update matrix stays in place (Symbolic::scatter_mode == 0, not a root)
  int32_t n_deferred_fronts = 0, deferred_max_m = 0;
  kh_spa_options opt;
  std::vector<Node> nodes;                       // insertion order
  std::unordered_map<int32_t, int32_t> index_of; // id -> position in nodes
  std::vector<Constraint> cons;
  std::unordered_multimap<uint64_t, int32_t> con_of;             // edge_key(a, b) -> position in cons (hashed: settle() re-keys
                                                                 // tens of thousands of constraints before a Compute)
  static uint64_t edge_key(int32_t a, int32_t b) {return (static_cast<uint64_t>(static_cast<uint32_t>(a)) << 32) | static_cast<uint32_t>(b);}
  // the constraint between a and b (in that order) that was added first, -1 = none
  int32_t first_constraint(int32_t a, int32_t b) const
  {
    int32_t best = -1;
    auto range = con_of.equal_range(edge_key(a, b));
    for (auto it = range.first; it != range.second; ++it) {if (best < 0 || it->second < best) {best = it->second;}}
    return best;
  }
  std::unordered_map<int32_t, std::vector<int32_t>> incident;   // node id -> positions in cons of its constraints (dead ones included until settle())
  int32_t n_dead = 0;                            // tombstones in nodes + cons (settle() compacts, order preserved)
  int32_t n_dead_nodes = 0;                      // ... of which nodes
  int32_t first_id = 0; bool has_first = false; bool was_constant_set = false;
  std::vector<int32_t> corr_ids; std::vector<double> corr_poses;
  bool topology_dirty = true;
  // cached problem
  Symbolic sym;
  std::vector<int32_t> free_of_node, node_of_free;
  int32_t fixed_index = -1;
  // signature of the topology the cached analysis belongs to (node count, edge endpoints in order, gauge): a graph that
  // is reset and re-added unchanged -- loadSerializedPoseGraph followed by Compute -- keeps its ordering, symbolic
  // factorisation and device index maps
  std::vector<int32_t> cached_ea, cached_eb; int32_t cached_n = -1, cached_fixed = -2;
  // supernodes of the last nested dissection by NODE ID (incremental re-analysis), the factorisation cost and size it had
  std::vector<std::vector<int32_t>> cached_sn_ids;
  int64_t cached_full_flops = 0; int32_t cached_full_nf = 0; int32_t reuse_count = 0; int32_t cached_full_levels = 0;
  int32_t last_analysis_incremental = 0;
  // device buffers
  DevBuf<int32_t> d_edge_a, d_edge_b, d_free_of_node, d_node_of_free, d_slot_contrib_ptr, d_slot_contrib,
    d_bsr_row_ptr, d_bsr_col, d_bsr_diag, d_node_contrib_ptr, d_node_contrib, d_front_m, d_front_ns,
    d_front_first, d_rows_ptr, d_rows, d_child_ptr, d_child_list, d_relpos_ptr, d_relpos, d_slot_ld,
    d_elim_of_free, d_free_of_elim, d_level_fronts, d_fail, d_sync;
  DevBuf<int64_t> d_front_off, d_slot_dest, d_winv_off;
  DevBuf<double> d_winv;                       // L11^-T of every front (level pipeline, round 3)
  DevBuf<FrontDesc> d_desc;
  uint8_t * h_upload = nullptr; size_t h_upload_cap = 0;     // pinned staging of an analysis' uploads (one block, ~30 copies out of it)
  DevBuf<int32_t> d_cinv;
  DevBuf<char> d_pack;                         // the analysis' arrays (the DevBufs the upload block of prepare_problem lists are its slices)
  DevBuf<double> d_fronts_b;                   // buffer B of the fronts (Symbolic::scatter_mode)
  DevBuf<double> d_edge_z, d_edge_u, d_edge_lin, d_edge_cost, d_Hg, d_fronts, d_x, d_cand, d_scale,
    d_diag, d_rhs, d_step, d_delta, d_scal;
  double * h_scal = nullptr; int32_t * h_fail = nullptr;
  // host-coherent block the last kernel of an iteration writes its scalars to, and the flag it raises behind them
  double * h_res = nullptr; int32_t * h_res_flag = nullptr; int32_t res_seq = 0;
  // trace of the last Compute(): one row per LM iteration, see kh_spa_iteration_log
  std::vector<std::array<double, 8>> iter_log;
  int32_t n_slots = 0;
  std::vector<int32_t> level_offsets, level_max_m, level_max_ns;
  // per level: the fronts behind the first level_split[l] of it go through k_front_update (they fit one workgroup's LDS), with
  // level_fused_lds[l] bytes of it; level_split[l] = the level's size: none
  std::vector<int32_t> level_split; std::vector<size_t> level_fused_lds;
  // per level: the most rows any of its fronts has below the pivot block (0: the root -- no update launches)
  std::vector<int32_t> level_max_nu;
  DevBuf<double> d_upd, d_fsb, d_partial;
  DevBuf<double> d_Hg_alt, d_best;          // normal equations at the candidate point (speculative); minimum-cost iterate
  // multi-GPU: edge-block sharded linearisation, H and g summed by the caller's collective
  int32_t shard_rank = 0, shard_world = 1;
  kh_allreduce_fn allreduce = nullptr; void * allreduce_user = nullptr;
  kh_comm * comm = nullptr;                    // RCCL communicator of kh_spa_set_comm: the all-reduce happens in the library
  // measurement: event pairs around the phases of an LM iteration (kMaxTimed iterations are timed, the rest only counted)
  static constexpr int kMaxTimed = 64;
  hipEvent_t ev_phase[kMaxTimed][4] = {};     // factor begin, factor end = backward begin, backward end, (spare)
  hipEvent_t ev_lin[2 * kMaxTimed + 2][2] = {};
  double last_symbolic_ms = 0.0;
  int32_t debug_flags = 0;                     // kh_spa_set_debug
};

// drops the tombstones; positions in `nodes` / `cons` and both maps are final again afterwards.  Every entry point that
// hands out positions, counts or the arrays themselves settles first.
static void settle(kh_spa * s)
{
  if (!s || s->n_dead == 0) {return;}
  s->cons.erase(std::remove_if(s->cons.begin(), s->cons.end(), [](const Constraint & c) {return c.dead != 0;}), s->cons.end());
  s->nodes.erase(std::remove_if(s->nodes.begin(), s->nodes.end(), [](const Node & n) {return n.dead != 0;}), s->nodes.end());
  s->con_of.clear();
  s->con_of.reserve(s->cons.size());
  s->incident.clear();
  for (size_t k = 0; k < s->cons.size(); ++k) {
    s->con_of.insert({kh_spa::edge_key(s->cons[k].a, s->cons[k].b), static_cast<int32_t>(k)});
    s->incident[s->cons[k].a].push_back(static_cast<int32_t>(k));
    s->incident[s->cons[k].b].push_back(static_cast<int32_t>(k));
  }
  s->index_of.clear();
  for (size_t k = 0; k < s->nodes.size(); ++k) {s->index_of[s->nodes[k].id] = static_cast<int32_t>(k);}
  s->n_dead = 0;
  s->n_dead_nodes = 0;
}

static void bury_constraint(kh_spa * s, int32_t k)
{
  Constraint & c = s->cons[k];
  if (c.dead) {return;}
  auto range = s->con_of.equal_range(kh_spa::edge_key(c.a, c.b));
  for (auto it = range.first; it != range.second; ++it) {
    if (it->second == k) {s->con_of.erase(it); break;}
  }
  c.dead = 1;
  ++s->n_dead;
  s->topology_dirty = true;
}

namespace kh
{

// ---- exact host pieces ------------------------------------------------------------------------
static void matrix3_inverse(const double * m, double * inv)    // Karto.h:2533-2577 (row-major 3x3)
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

// AddConstraint, ceres_solver.cpp:364-376: information = symmetrised inverse (its upper triangle, row-major:
// 00 01 02 11 12 22); U = llt().matrixU()
static void information_from_covariance(const double * cov, double * omega)
{
  double p[9];
  matrix3_inverse(cov, p);
  omega[0] = p[0]; omega[1] = p[1]; omega[2] = p[2]; omega[3] = p[4]; omega[4] = p[5]; omega[5] = p[8];
}

static void sqrt_information(const double * omega, double * U)
{
  const double a00 = omega[0], a01 = omega[1], a02 = omega[2], a11 = omega[3], a12 = omega[4], a22 = omega[5];
  const double l00 = std::sqrt(a00);
  const double l10 = a01 / l00, l20 = a02 / l00;
  const double l11 = std::sqrt(a11 - l10 * l10);
  const double l21 = (a12 - l20 * l10) / l11;
  const double l22 = std::sqrt(a22 - l20 * l20 - l21 * l21);
  U[0] = l00; U[1] = l10; U[2] = l20;
  U[3] = 0.0; U[4] = l11; U[5] = l21;
  U[6] = 0.0; U[7] = 0.0; U[8] = l22;
}

static double karto_normalize_angle(double angle)   // Math.h:181-202
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

// ---- problem setup on the device -----------------------------------------------------------------
static int prepare_problem(kh_spa * s, SpaDev & dev, bool & has_work)
{
  const int32_t N = static_cast<int32_t>(s->nodes.size());
  const int32_t E = static_cast<int32_t>(s->cons.size());
  has_work = false;
  s->last_symbolic_ms = 0.0;
  const auto t_prep0 = std::chrono::steady_clock::now();
  // gauge: first inserted node is constant once it has parameter blocks (ceres_solver.cpp:228-241)
  std::vector<uint8_t> used(N, 0);
  std::vector<int32_t> ea(E), eb(E);
  {
    // node ids are scan ids: small and dense, so the 2 E lookups go through a table (a hash lookup each was 2 ms of every
    // Compute on a 57 000-edge graph); sparse or negative ids keep the map
    int32_t lo = 0, hi = -1;
    for (const Node & nd : s->nodes) {lo = std::min(lo, nd.id); hi = std::max(hi, nd.id);}
    std::vector<int32_t> pos_of_id;
    if (lo >= 0 && hi >= 0 && static_cast<int64_t>(hi) < 8ll * N + 4096) {
      pos_of_id.assign(static_cast<size_t>(hi) + 1, -1);
      for (int32_t i = 0; i < N; ++i) {pos_of_id[s->nodes[i].id] = i;}
    }
    for (int32_t e = 0; e < E; ++e) {
      if (!pos_of_id.empty()) {
        ea[e] = pos_of_id[s->cons[e].a]; eb[e] = pos_of_id[s->cons[e].b];
      } else {
        ea[e] = s->index_of.at(s->cons[e].a); eb[e] = s->index_of.at(s->cons[e].b);
      }
      used[ea[e]] = 1; used[eb[e]] = 1;
    }
  }
  if (!s->was_constant_set && s->has_first) {
    auto it = s->index_of.find(s->first_id);
    if (it != s->index_of.end() && used[it->second]) {s->was_constant_set = true;}
  }
  int32_t fixed = -1;
  if (s->was_constant_set) {
    auto it = s->index_of.find(s->first_id);
    if (it != s->index_of.end()) {fixed = it->second;}
  }
  if (s->topology_dirty && fixed == s->cached_fixed && N == s->cached_n && ea == s->cached_ea && eb == s->cached_eb &&
    !s->node_of_free.empty())
  {
    s->topology_dirty = false;          // same nodes, same edges in the same order, same gauge: the cached analysis holds
    s->fixed_index = fixed;
  }
  if (s->topology_dirty || fixed != s->fixed_index) {
    s->fixed_index = fixed;
    s->free_of_node.assign(N, -1);
    s->node_of_free.clear();
    for (int32_t i = 0; i < N; ++i) {
      if (used[i] && i != fixed) {s->free_of_node[i] = static_cast<int32_t>(s->node_of_free.size()); s->node_of_free.push_back(i);}
    }
    const int32_t nf = static_cast<int32_t>(s->node_of_free.size());
    if (nf == 0 || E == 0) {s->topology_dirty = true; return KH_OK;}
    // adjacency + BSR pattern over the free nodes (flat CSR arrays, counting passes: no per-row vectors)
    std::vector<int32_t> deg(nf + 1, 0);
    for (int32_t e = 0; e < E; ++e) {
      const int32_t fa = s->free_of_node[ea[e]], fb = s->free_of_node[eb[e]];
      if (fa >= 0 && fb >= 0 && fa != fb) {++deg[fa + 1]; ++deg[fb + 1];}
    }
    for (int32_t i = 0; i < nf; ++i) {deg[i + 1] += deg[i];}
    std::vector<int32_t> nbr(deg[nf]), fill(deg.begin(), deg.end() - 1);
    for (int32_t e = 0; e < E; ++e) {
      const int32_t fa = s->free_of_node[ea[e]], fb = s->free_of_node[eb[e]];
      if (fa >= 0 && fb >= 0 && fa != fb) {nbr[fill[fa]++] = fb; nbr[fill[fb]++] = fa;}
    }
    // sorted, duplicate-free neighbour lists (compact) and the BSR pattern (a row's neighbours with the diagonal block in its
    // sorted place).  Rows are independent: blocks of rows on the host pool sort their lists in place, a serial prefix sum
    // places the rows, the blocks write them out.
    std::vector<int32_t> adj_ptr(nf + 1, 0), adj_idx, row_ptr(nf + 1, 0), col, diag(nf), slot_row, ucount(nf, 0);
    constexpr int32_t kRowBlock = 1024;
    const size_t row_blocks = static_cast<size_t>((nf + kRowBlock - 1) / kRowBlock);
    host_parallel_for(row_blocks, [&](size_t blk) {
      const int32_t i_end = std::min(nf, static_cast<int32_t>(blk + 1) * kRowBlock);
      for (int32_t i = static_cast<int32_t>(blk) * kRowBlock; i < i_end; ++i) {
        int32_t * b = nbr.data() + deg[i], * e2 = nbr.data() + deg[i + 1];
        std::sort(b, e2);
        ucount[i] = static_cast<int32_t>(std::unique(b, e2) - b);
      }
    });
    for (int32_t i = 0; i < nf; ++i) {adj_ptr[i + 1] = adj_ptr[i] + ucount[i]; row_ptr[i + 1] = row_ptr[i] + ucount[i] + 1;}
    adj_idx.resize(adj_ptr[nf]); col.resize(row_ptr[nf]); slot_row.resize(row_ptr[nf]);
    host_parallel_for(row_blocks, [&](size_t blk) {
      const int32_t i_end = std::min(nf, static_cast<int32_t>(blk + 1) * kRowBlock);
      for (int32_t i = static_cast<int32_t>(blk) * kRowBlock; i < i_end; ++i) {
        const int32_t * b = nbr.data() + deg[i], * e2 = b + ucount[i];
        std::copy(b, e2, adj_idx.begin() + adj_ptr[i]);
        int32_t at = row_ptr[i];
        bool placed = false;
        for (const int32_t * q = b; q < e2; ++q) {
          if (!placed && *q > i) {diag[i] = at; col[at] = i; slot_row[at] = i; ++at; placed = true;}
          col[at] = *q; slot_row[at] = i; ++at;
        }
        if (!placed) {diag[i] = at; col[at] = i; slot_row[at] = i; ++at;}
      }
    });
    const int32_t n_slots = static_cast<int32_t>(col.size());
    // Ordering + fronts on a thread of their own, beside the contribution lists below (both only read the adjacency).
    // INCREMENTAL re-analysis: a loop closure adds a few dozen nodes and edges to a graph that was dissected milliseconds
    // ago (and lifelong mode removes a few).  The supernodes of the last dissection, kept by node id, are reused -- the
    // nodes that have left drop out, the new ones become leading leaf supernodes (eliminated first: no extra levels; their
    // old neighbours become mutually adjacent, which the structure pass accounts for like any fill) -- and only the structure
    // is rebuilt.  A full dissection again when the new nodes pass a quarter of the graph, when the factorisation's flops have tripled relative to the graph's growth,
    // or after kMaxReuse re-analyses.  kh_spa_reset() forgets the supernodes (a reloaded graph is analysed from scratch).
    const auto t_sym0 = std::chrono::steady_clock::now();
    SymbolicOptions sopt;
    // the independent subsets of a dissection level run on the library's persistent host pool (KH_SPA_ND_SERIAL=1: in line)
    static const bool nd_serial = std::getenv("KH_SPA_ND_SERIAL") != nullptr;
    if (!nd_serial) {sopt.parallel_for = [](size_t n, const std::function<void(size_t)> & fn) {host_parallel_for(n, fn);};}
    if (const char * e = std::getenv("KH_SPA_LEAF")) {sopt.leaf_nodes = std::max(1, std::atoi(e));}
    if (const char * e = std::getenv("KH_SPA_PMAX")) {sopt.max_pivot_nodes = std::min(42, std::max(1, std::atoi(e)));}
    if (const char * e = std::getenv("KH_SPA_CANDS")) {sopt.separator_candidates = std::max(1, std::atoi(e));}
    static const bool incremental_on = !(std::getenv("KH_SPA_INCREMENTAL") && std::atoi(std::getenv("KH_SPA_INCREMENTAL")) == 0);
    int sym_rc = KH_OK;
    bool sym_incremental = false;
    std::string sym_error;
    auto analyse = [&]() {
      constexpr int kMaxReuse = 24;
      std::vector<std::vector<int32_t>> sn;
      if (incremental_on && !s->cached_sn_ids.empty() && s->reuse_count < kMaxReuse) {
        // node id -> free index of this problem: a table when the ids are dense (a mapper's scan ids are), a hash map otherwise
        int32_t id_lo = INT32_MAX, id_hi = INT32_MIN;
        for (int32_t f = 0; f < nf; ++f) {const int32_t id = s->nodes[s->node_of_free[f]].id; id_lo = std::min(id_lo, id); id_hi = std::max(id_hi, id);}
        const bool dense_ids = nf > 0 && static_cast<int64_t>(id_hi) - id_lo < 8 * static_cast<int64_t>(nf) + 1024;
        std::vector<int32_t> free_of_dense;
        std::unordered_map<int32_t, int32_t> free_of_id;
        if (dense_ids) {
          free_of_dense.assign(static_cast<size_t>(id_hi - id_lo) + 1, -1);
          for (int32_t f = 0; f < nf; ++f) {free_of_dense[static_cast<size_t>(s->nodes[s->node_of_free[f]].id - id_lo)] = f;}
        } else {
          free_of_id.reserve(static_cast<size_t>(nf) * 2);
          for (int32_t f = 0; f < nf; ++f) {free_of_id[s->nodes[s->node_of_free[f]].id] = f;}
        }
        auto free_index = [&](int32_t id) -> int32_t {
          if (dense_ids) {return (id >= id_lo && id <= id_hi) ? free_of_dense[static_cast<size_t>(id - id_lo)] : -1;}
          const auto it = free_of_id.find(id);
          return it != free_of_id.end() ? it->second : -1;
        };
        std::vector<uint8_t> placed(nf, 0);
        int32_t kept = 0;
        sn.reserve(s->cached_sn_ids.size() + 1);
        for (const auto & ids : s->cached_sn_ids) {
          std::vector<int32_t> g;
          g.reserve(ids.size());
          for (int32_t id : ids) {
            const int32_t f = free_index(id);
            if (f >= 0) {g.push_back(f); placed[f] = 1; ++kept;}
          }
          if (!g.empty()) {std::sort(g.begin(), g.end()); sn.push_back(std::move(g));}
        }
        if (nf - kept <= std::max(64, nf / 4)) {
          std::vector<int32_t> fresh;
          for (int32_t f = 0; f < nf; ++f) {if (!placed[f]) {fresh.push_back(f);}}
          if (!fresh.empty()) {sn.insert(sn.begin(), std::move(fresh));}          // build_structure splits it into a chain when long
          sym_rc = build_structure(s->sym, nf, adj_ptr, adj_idx, sopt, sn);
          // fill guard: the factorisation may cost three times the last full dissection's, scaled by how much the graph has
          // grown since (flops grow at least linearly with the free nodes).  Three, not one and a half: the leading leaves of a
          // closure's few dozen new nodes add little fill, but a bound of 1.5 sent most closures of the 50 000-scan replay back
          // to a full dissection (5 ms each: the replay's solver time went from 4.1 to 7.5 s)
          const double allowed = 3.0 * static_cast<double>(s->cached_full_flops) * static_cast<double>(nf) / static_cast<double>(std::max(1, s->cached_full_nf));
          // level guard (round 6): the leading leaf's update matrix reaches into every supernode its nodes touch and strings those up
          // as ancestors of one another -- on a connected graph (the non-lifelong replay) every re-analysis added one to three levels of
          // ONE front each to the top of the tree (8 levels after a full dissection, 26 sixteen closures later), and a level is three
          // launches of 10-27 us per factorisation whatever it holds.  Back to a full dissection once the tree is extra_levels taller
          // than the last one left it.  Measured (solver ms of the 3000-scan non-lifelong / the 20 000-scan lifelong replay): no guard
          // 499 / 810, 5 levels 418 / 647, 3: 384 / 642, 2: 364-377 / 508, 1: 366 / 518, 0: 370 / 602, never incremental 358 / 621.
          // On the 50 000-scan lifelong replay (12 000 free nodes in thousands of components: a dissection costs 5-7 ms, the tree
          // grows slowly) 2 / 6 / no guard gave 2772 / 2576 / 2606 ms (medians of three alternating runs): 6 from 6000 free nodes on.
          // (KH_SPA_EXTRA_LEVELS is read at every analysis: tests/test_spa_gpu.py switches the guard off around one Compute().)
          const char * guard_env = std::getenv("KH_SPA_EXTRA_LEVELS");
          const int extra_levels = guard_env ? std::atoi(guard_env) : (nf <= 6000 ? 2 : 6);
          const bool levels_ok = static_cast<int>(s->sym.levels.size()) <= s->cached_full_levels + extra_levels;
          if (sym_rc == KH_OK && static_cast<double>(s->sym.factor_flops) <= allowed && levels_ok) {
            sym_incremental = true;
            return;
          }
        }
      }
      std::vector<std::vector<int32_t>> dissected;
      sym_rc = nested_dissection(nf, adj_ptr, adj_idx, sopt, dissected);
      if (sym_rc == KH_OK) {
        s->cached_sn_ids.assign(dissected.size(), {});
        for (size_t k = 0; k < dissected.size(); ++k) {
          for (int32_t f : dissected[k]) {s->cached_sn_ids[k].push_back(s->nodes[s->node_of_free[f]].id);}
        }
        sym_rc = build_structure(s->sym, nf, adj_ptr, adj_idx, sopt, dissected);
        s->cached_full_flops = s->sym.factor_flops; s->cached_full_nf = nf; s->reuse_count = 0;
        s->cached_full_levels = static_cast<int32_t>(s->sym.levels.size());
      }
      if (sym_rc != KH_OK) {sym_error = kh_last_error();}
    };
    // (a thread of its own only where there is something to win: the lists below take 0.6 ms on the 10 000-node graph and 0.08 ms on
    // a mapper's 2000 nodes, creating and joining a thread 0.1 ms)
    const bool analysis_beside = nf >= 4000;
    std::thread sym_thread;
    if (analysis_beside) {sym_thread = std::thread(analyse);}
    struct Joiner {std::thread & t; ~Joiner() {if (t.joinable()) {t.join();}}} sym_joiner{sym_thread};
    auto slot_of = [&](int32_t i, int32_t j) {
      const auto b = col.begin() + row_ptr[i], e2 = col.begin() + row_ptr[i + 1];
      return static_cast<int32_t>(std::lower_bound(b, e2, j) - col.begin());
    };
    // contribution lists of the gather kernels: two counting passes in edge order (same order as before:
    // ascending edge index within every slot / node)
    std::vector<int32_t> scp(n_slots + 1, 0), ncp(nf + 1, 0);
    std::vector<int32_t> e_slots(static_cast<size_t>(E) * 4, -1);
    for (int32_t e = 0; e < E; ++e) {
      const int32_t fa = s->free_of_node[ea[e]], fb = s->free_of_node[eb[e]];
      if (fa >= 0) {e_slots[4 * e + 0] = diag[fa]; ++scp[diag[fa] + 1]; ++ncp[fa + 1];}
      if (fb >= 0) {e_slots[4 * e + 1] = diag[fb]; ++scp[diag[fb] + 1]; ++ncp[fb + 1];}
      if (fa >= 0 && fb >= 0 && fa != fb) {
        e_slots[4 * e + 2] = slot_of(fa, fb); e_slots[4 * e + 3] = slot_of(fb, fa);
        ++scp[e_slots[4 * e + 2] + 1]; ++scp[e_slots[4 * e + 3] + 1];
      }
    }
    for (int32_t k = 0; k < n_slots; ++k) {scp[k + 1] += scp[k];}
    for (int32_t i = 0; i < nf; ++i) {ncp[i + 1] += ncp[i];}
    std::vector<int32_t> scl(scp[n_slots]), ncl(ncp[nf]), sfill(scp.begin(), scp.end() - 1), nfill(ncp.begin(), ncp.end() - 1);
    for (int32_t e = 0; e < E; ++e) {
      const int32_t fa = s->free_of_node[ea[e]], fb = s->free_of_node[eb[e]];
      for (int kind = 0; kind < 4; ++kind) {
        const int32_t sl = e_slots[4 * e + kind];
        if (sl >= 0) {scl[sfill[sl]++] = e * 4 + kind;}
      }
      if (fa >= 0) {ncl[nfill[fa]++] = e * 2 + 0;}
      if (fb >= 0) {ncl[nfill[fb]++] = e * 2 + 1;}
    }
    const auto t_lists = std::chrono::steady_clock::now();
    if (analysis_beside) {sym_thread.join();} else {analyse();}
    if (sym_rc) {set_error(sym_error); return sym_rc;}
    if (sym_incremental) {++s->reuse_count;}
    s->last_analysis_incremental = sym_incremental ? 1 : 0;
    int rc = KH_OK;
    if (std::getenv("KH_SPA_DEBUG")) {
      std::fprintf(stderr, "[kh_spa] host: adjacency %.2f ms, lists %.2f ms beside the %s analysis (%.2f ms)\n",
        std::chrono::duration<double, std::milli>(t_sym0 - t_prep0).count(),
        std::chrono::duration<double, std::milli>(t_lists - t_sym0).count(), sym_incremental ? "incremental" : "full",
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sym0).count());
    }
    if (std::getenv("KH_SPA_DEBUG")) {
      int32_t max_m = 0, max_ns = 0;
      for (int32_t k = 0; k < s->sym.n_fronts; ++k) {max_m = std::max(max_m, s->sym.front_m[k]); max_ns = std::max(max_ns, s->sym.front_ns[k]);}
      std::fprintf(stderr, "[kh_spa] free nodes %d, fronts %d, levels %zu, max front m %d, max ns %d, front storage %.1f MB, nnz(L) %lld\n",
        nf, s->sym.n_fronts, s->sym.levels.size(), max_m, max_ns, s->sym.fronts_size * 8.0 / 1e6, static_cast<long long>(s->sym.nnz_factor));
      for (size_t l = 0; l < s->sym.levels.size(); ++l) {
        int32_t mm = 0; int64_t work = 0;
        for (int32_t k : s->sym.levels[l]) {mm = std::max(mm, s->sym.front_m[k]); work += static_cast<int64_t>(s->sym.front_m[k]) * s->sym.front_m[k] * s->sym.front_ns[k];}
        std::fprintf(stderr, "[kh_spa]   level %zu: %zu fronts, max m %d, flops~%.2fM\n", l, s->sym.levels[l].size(), mm, work / 1e6);
      }
    }
    const Symbolic & sym = s->sym;
    std::vector<int64_t> slot_dest(n_slots, -1);
    std::vector<int32_t> slot_ld(n_slots, 0);
    {
      // where every block of the pattern lands in its front: independent per slot, blocks of slots on the host pool
      constexpr int32_t kSlotBlock = 8192;
      std::atomic<int> outside{0};
      host_parallel_for(static_cast<size_t>((n_slots + kSlotBlock - 1) / kSlotBlock), [&](size_t blk) {
        const int32_t k_end = std::min(n_slots, static_cast<int32_t>(blk + 1) * kSlotBlock);
        for (int32_t k = static_cast<int32_t>(blk) * kSlotBlock; k < k_end; ++k) {
          const int32_t ei = sym.elim_of_free[slot_row[k]], ej = sym.elim_of_free[col[k]];
          if (ei < ej) {continue;}
          const int32_t f = sym.sn_of_elim[ej];
          const int32_t ncols = sym.front_ns[f] / 3;
          const int32_t colpos = ej - sym.front_first[f];
          int32_t rowpos;
          if (ei < sym.front_first[f] + ncols) {
            rowpos = ei - sym.front_first[f];
          } else {
            const auto b = sym.rows.begin() + sym.rows_ptr[f], e2 = sym.rows.begin() + sym.rows_ptr[f + 1];
            auto it = std::lower_bound(b, e2, ei);
            if (it == e2 || *it != ei) {outside.store(1); continue;}
            rowpos = ncols + static_cast<int32_t>(it - b);
          }
          slot_dest[k] = sym.front_off[f] + 3 * rowpos + static_cast<int64_t>(3 * colpos) * sym.front_m[f];
          slot_ld[k] = sym.front_m[f];
        }
      });
      if (outside.load()) {set_error("symbolic: matrix entry outside its front"); return KH_ERR_SOLVER;}
    }
    // level lists, concatenated
    std::vector<int32_t> level_fronts;
    s->level_offsets.assign(1, 0);
    s->level_max_m.clear(); s->level_max_ns.clear(); s->level_split.clear(); s->level_fused_lds.clear(); s->level_max_nu.clear();
    for (auto & lv : sym.levels) {
      level_fronts.insert(level_fronts.end(), lv.begin(), lv.end());
      s->level_offsets.push_back(static_cast<int32_t>(level_fronts.size()));
      int32_t mm = 0, mns = 0;
      for (int32_t k : lv) {mm = std::max(mm, sym.front_m[k]); mns = std::max(mns, sym.front_ns[k]);}
      s->level_max_m.push_back(mm); s->level_max_ns.push_back(mns);
      // the fronts of a level are numbered largest first: the ones that do not fit k_front_update are (nearly) a prefix
      static const int fuse_min = std::getenv("KH_SPA_FUSE_MIN") ? std::atoi(std::getenv("KH_SPA_FUSE_MIN")) : 256;
      int32_t split = 0;
      size_t lds = 0;
      for (size_t q = 0; q < lv.size(); ++q) {
        const size_t b = spa_front_update_lds(sym.front_m[lv[q]], sym.front_ns[lv[q]]);
        if (b == 0 && sym.front_m[lv[q]] > sym.front_ns[lv[q]]) {split = static_cast<int32_t>(q) + 1; lds = 0;} else {lds = std::max(lds, b);}
      }
      if (static_cast<int32_t>(lv.size()) - split < fuse_min) {split = static_cast<int32_t>(lv.size());}
      s->level_split.push_back(split); s->level_fused_lds.push_back(lds);
      int32_t mnu = 0;
      for (int32_t k : lv) {mnu = std::max(mnu, sym.front_m[k] - sym.front_ns[k]);}
      s->level_max_nu.push_back(mnu);
    }
    // uploads
    hipStream_t st = s->stream;
    std::vector<int32_t> col_and_row = col;
    col_and_row.insert(col_and_row.end(), slot_row.begin(), slot_row.end());
    int r2 = 0;
    // Every array is copied into ONE pinned block and travels to ONE device block in ONE copy; the device buffers of the arrays are
    // slices of that block (round 5: one pinned block, ~35 asynchronous copies -- 3-4 us of the caller's time each, on every
    // closure of a mapper; before that ~35 copies out of pageable vectors, each staged and waited for on its own).
    struct Pending {std::function<void(char *)> bind; const void * src; size_t bytes; size_t off;};
    std::vector<Pending> pending;
    size_t pack_bytes = 0;
    auto up = [&](auto & buf, const auto & vec) {
      typedef typename std::remove_reference<decltype(buf)>::type::value_type T;
      const size_t n = std::max<size_t>(vec.size(), 1);
      auto * bp = &buf;
      pending.push_back({[bp, n](char * base) {bp->alias(reinterpret_cast<T *>(base), n);}, vec.empty() ? nullptr : vec.data(), vec.size() * sizeof(T), pack_bytes});
      pack_bytes += (n * sizeof(T) + 63) & ~static_cast<size_t>(63);
    };
    up(s->d_edge_a, ea); up(s->d_edge_b, eb);
    up(s->d_free_of_node, s->free_of_node); up(s->d_node_of_free, s->node_of_free);
    up(s->d_slot_contrib_ptr, scp); up(s->d_slot_contrib, scl);
    up(s->d_bsr_row_ptr, row_ptr); up(s->d_bsr_col, col_and_row); up(s->d_bsr_diag, diag);
    up(s->d_node_contrib_ptr, ncp); up(s->d_node_contrib, ncl);
    up(s->d_front_off, sym.front_off); up(s->d_front_m, sym.front_m);
    up(s->d_front_ns, sym.front_ns); up(s->d_front_first, sym.front_first);
    up(s->d_rows_ptr, sym.rows_ptr); up(s->d_rows, sym.rows);
    up(s->d_child_ptr, sym.child_ptr); up(s->d_child_list, sym.child_list);
    up(s->d_relpos_ptr, sym.relpos_ptr); up(s->d_relpos, sym.relpos);
    up(s->d_slot_dest, slot_dest); up(s->d_slot_ld, slot_ld);
    up(s->d_elim_of_free, sym.elim_of_free); up(s->d_free_of_elim, sym.free_of_elim);
    up(s->d_level_fronts, level_fronts);
    std::vector<int32_t> deferred;
    s->deferred_max_m = 0;
    for (int32_t k = 0; k < sym.n_fronts; ++k) {
      if (sym.scatter_mode[k] == 0 && sym.parent[k] >= 0 && sym.front_m[k] > sym.front_ns[k]) {
        deferred.push_back(k); s->deferred_max_m = std::max(s->deferred_max_m, sym.front_m[k]);
      }
    }
    s->n_deferred_fronts = static_cast<int32_t>(deferred.size());
    if (deferred.empty()) {deferred.push_back(0);}
    up(s->d_deferred, deferred);
    up(s->d_winv_off, sym.winv_off);
    std::vector<FrontDesc> desc(sym.n_fronts);
    for (int32_t k = 0; k < sym.n_fronts; ++k) {
      FrontDesc & fd = desc[k];
      std::memset(&fd, 0, sizeof(fd));
      fd.off = sym.front_off[k]; fd.woff = sym.winv_off[k]; fd.m = sym.front_m[k]; fd.ns = sym.front_ns[k]; fd.first = sym.front_first[k];
      fd.rows_ptr = sym.rows_ptr[k]; fd.child_ptr = sym.child_ptr[k]; fd.child_end = sym.child_ptr[k + 1];
      fd.relpos_ptr = sym.relpos_ptr[k]; fd.parent = sym.parent[k]; fd.cinv_ptr = sym.cinv_ptr[k];
      fd.flags = sym.scatter_mode[k] | (sym.has_b[k] ? 4 : 0) | (sym.n_deferred[k] << 8);
      for (int32_t q = 0; q < 3 && fd.child_ptr + q < fd.child_end; ++q) {
        const int32_t c = sym.child_list[fd.child_ptr + q];
        fd.ch[q].off = sym.front_off[c]; fd.ch[q].m = sym.front_m[c]; fd.ch[q].ns = sym.front_ns[c]; fd.ch[q].rows_ptr = sym.rows_ptr[c];
      }
    }
    up(s->d_cinv, sym.cinv);
    up(s->d_desc, desc);
    {
      const size_t total = std::max<size_t>(pack_bytes, 64);
      if (total > s->h_upload_cap) {
        KS_HIP(hipStreamSynchronize(st));          // (an earlier copy may still read the old block)
        if (s->h_upload) {KS_HIP(hipHostFree(s->h_upload)); s->h_upload = nullptr; s->h_upload_cap = 0;}
        const size_t cap = total + total / 2;
        KS_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_upload), cap, hipHostMallocDefault));
        s->h_upload_cap = cap;
      }
      if (total > s->d_pack.cap) {KS_HIP(hipStreamSynchronize(st));}       // (kernels of the last solve read slices of the old block)
      r2 |= s->d_pack.ensure(total);
      if (r2) {return KH_ERR_HIP;}
      for (const Pending & q : pending) {
        if (q.bytes) {std::memcpy(s->h_upload + q.off, q.src, q.bytes);}
        q.bind(s->d_pack.p + q.off);
      }
      if (hipMemcpyAsync(s->d_pack.p, s->h_upload, total, hipMemcpyHostToDevice, st) != hipSuccess) {
        (void)hipStreamSynchronize(st);
        set_error("hipMemcpyAsync: upload of the analysis");
        return KH_ERR_HIP;
      }
    }
    r2 |= s->d_winv.ensure(static_cast<size_t>(sym.winv_size) + 16);
    if (r2) {(void)hipStreamSynchronize(st); return KH_ERR_HIP;}
    s->n_slots = n_slots;
    const size_t scratch = std::max<size_t>(static_cast<size_t>(21) * E, static_cast<size_t>(9) * nf + 16);
    r2 |= s->d_edge_lin.ensure(scratch); r2 |= s->d_edge_cost.ensure(std::max(E, 1));
    // H and g share one buffer so that a sharded run sums them across ranks with ONE all-reduce
    r2 |= s->d_Hg.ensure(static_cast<size_t>(n_slots) * 9 + static_cast<size_t>(nf) * 3 + 8);
    r2 |= s->d_Hg_alt.ensure(static_cast<size_t>(n_slots) * 9 + static_cast<size_t>(nf) * 3 + 8);
    r2 |= s->d_fronts.ensure(static_cast<size_t>(sym.fronts_size) + 16);
    r2 |= s->d_fronts_b.ensure(static_cast<size_t>(sym.fronts_size) + 16);
    r2 |= s->d_scale.ensure(3 * nf); r2 |= s->d_diag.ensure(3 * nf); r2 |= s->d_rhs.ensure(3 * nf);
    r2 |= s->d_step.ensure(3 * nf); r2 |= s->d_delta.ensure(3 * nf);
    r2 |= s->d_fail.ensure(4);
    r2 |= s->d_sync.ensure(4 * static_cast<size_t>(sym.n_fronts) + 4);
    r2 |= s->d_upd.ensure(static_cast<size_t>(3) * sym.rows_ptr[sym.n_fronts] + 16);
    r2 |= s->d_partial.ensure(static_cast<size_t>(5) * ((4 * nf + 255) / 256) + (E + 255) / 256 + static_cast<size_t>(2) * ((3 * nf + 255) / 256) + 32);
    r2 |= s->d_fsb.ensure(static_cast<size_t>(3) * (static_cast<size_t>(nf) + sym.rows_ptr[sym.n_fronts]) + 16);
    if (r2) {(void)hipStreamSynchronize(st); return KH_ERR_HIP;}
    // the uploads above read pageable host vectors of this block: they must have landed before the block ends
    KS_HIP(hipStreamSynchronize(st));
    s->topology_dirty = false;
    s->cached_ea = ea; s->cached_eb = eb; s->cached_n = N; s->cached_fixed = fixed;
    s->last_symbolic_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep0).count();
  }
  const int32_t nf = static_cast<int32_t>(s->node_of_free.size());
  if (nf == 0 || E == 0) {return KH_OK;}
  // values that change between calls without touching the topology
  std::vector<double> z(static_cast<size_t>(E) * 3), u(static_cast<size_t>(E) * 9), x(static_cast<size_t>(N) * 3);
  for (int32_t e = 0; e < E; ++e) {
    std::copy(s->cons[e].z, s->cons[e].z + 3, z.begin() + 3 * e);
    std::copy(s->cons[e].u, s->cons[e].u + 9, u.begin() + 9 * e);
  }
  for (int32_t i = 0; i < N; ++i) {std::copy(s->nodes[i].pose, s->nodes[i].pose + 3, x.begin() + 3 * i);}
  int r3 = 0;
  r3 |= s->d_edge_z.upload(z, s->stream); r3 |= s->d_edge_u.upload(u, s->stream);
  r3 |= s->d_x.upload(x, s->stream); r3 |= s->d_cand.ensure(x.size()); r3 |= s->d_best.ensure(x.size()); r3 |= s->d_scal.ensure(32);
  if (r3) {return KH_ERR_HIP;}
  KS_HIP(hipStreamSynchronize(s->stream));   // the staging vectors above go out of scope
  if (std::getenv("KH_SPA_DEBUG")) {
    std::fprintf(stderr, "[kh_spa] host: prepare_problem total %.2f ms\n",
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_prep0).count());
  }

  const Symbolic & sym = s->sym;
  dev.n_nodes = N; dev.n_free = nf; dev.n_edges = E;
  dev.edge_a = s->d_edge_a.p; dev.edge_b = s->d_edge_b.p; dev.edge_z = s->d_edge_z.p; dev.edge_u = s->d_edge_u.p;
  dev.free_of_node = s->d_free_of_node.p; dev.node_of_free = s->d_node_of_free.p;
  dev.edge_lin = s->d_edge_lin.p; dev.edge_cost = s->d_edge_cost.p;
  dev.n_slots = s->n_slots; dev.slot_contrib_ptr = s->d_slot_contrib_ptr.p; dev.slot_contrib = s->d_slot_contrib.p;
  dev.bsr_row_ptr = s->d_bsr_row_ptr.p; dev.bsr_col = s->d_bsr_col.p; dev.bsr_diag_slot = s->d_bsr_diag.p;
  dev.H = s->d_Hg.p; dev.node_contrib_ptr = s->d_node_contrib_ptr.p; dev.node_contrib = s->d_node_contrib.p;
  dev.g = s->d_Hg.p + static_cast<size_t>(s->n_slots) * 9;
  dev.n_fronts = sym.n_fronts; dev.front_off = s->d_front_off.p; dev.front_m = s->d_front_m.p; dev.front_ns = s->d_front_ns.p;
  dev.front_first = s->d_front_first.p; dev.front_rows_ptr = s->d_rows_ptr.p; dev.front_rows = s->d_rows.p;
  dev.child_ptr = s->d_child_ptr.p; dev.child_list = s->d_child_list.p; dev.relpos_ptr = s->d_relpos_ptr.p; dev.relpos = s->d_relpos.p;
  dev.slot_dest = s->d_slot_dest.p; dev.slot_ld = s->d_slot_ld.p;
  dev.elim_of_free = s->d_elim_of_free.p; dev.free_of_elim = s->d_free_of_elim.p;
  dev.fronts = s->d_fronts.p; dev.fronts_size = sym.fronts_size; dev.fronts_b = s->d_fronts_b.p;
  dev.winv = s->d_winv.p; dev.winv_off = s->d_winv_off.p; dev.desc = s->d_desc.p; dev.cinv = s->d_cinv.p;
  has_work = true;
  return KH_OK;
}

}  // namespace kh

// =============================================================================================
extern "C" {

void kh_spa_options_default(kh_spa_options * o)
{
  // ceres_solver.cpp:157-186; max_num_iterations is left at the Ceres default
  o->max_num_iterations = 50;
  o->function_tolerance = 1e-3; o->gradient_tolerance = 1e-6; o->parameter_tolerance = 1e-3;
  o->min_relative_decrease = 1e-3;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e8; o->min_trust_region_radius = 1e-16;
  o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 3; o->use_nonmonotonic_steps = 1;
  o->max_consecutive_nonmonotonic_steps = 3; o->jacobi_scaling = 1;
  o->loss_function = KH_LOSS_NONE; o->loss_scale = 0.7;
}

int kh_spa_create(int32_t device, kh_spa ** out)
{
  if (!out) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  kh_spa * s = new kh_spa();
  s->device = device;
  kh_spa_options_default(&s->opt);
  KS_HIP(hipSetDevice(device));
  KS_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  KS_HIP(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
  for (auto & e : s->ev_level) {KS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));}
  KS_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_scal), sizeof(double) * 32, hipHostMallocDefault));
  KS_HIP(hipHostMalloc(reinterpret_cast<void **>(&s->h_fail), sizeof(int32_t) * 4, hipHostMallocDefault));
  // (a platform without host-coherent mappings keeps the copies + stream drain)
  if (hipHostMalloc(reinterpret_cast<void **>(&s->h_res), sizeof(double) * 16 + 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    s->h_res_flag = reinterpret_cast<int32_t *>(s->h_res + 16);
    s->h_res_flag[0] = 0;
  } else {
    (void)hipGetLastError();
    s->h_res = nullptr;
  }
  for (auto & row : s->ev_phase) {for (auto & e : row) {KS_HIP(hipEventCreate(&e));}}
  for (auto & row : s->ev_lin) {for (auto & e : row) {KS_HIP(hipEventCreate(&e));}}
  *out = s;
  return KH_OK;
}

void kh_spa_destroy(kh_spa * s)
{
  if (!s) {return;}
  (void)hipSetDevice(s->device);
  if (s->stream) {(void)hipStreamSynchronize(s->stream);}
  s->d_edge_a.release(); s->d_edge_b.release(); s->d_free_of_node.release(); s->d_node_of_free.release();
  s->d_slot_contrib_ptr.release(); s->d_slot_contrib.release(); s->d_bsr_row_ptr.release(); s->d_bsr_col.release();
  s->d_bsr_diag.release(); s->d_node_contrib_ptr.release(); s->d_node_contrib.release(); s->d_front_m.release();
  s->d_front_ns.release(); s->d_front_first.release(); s->d_rows_ptr.release(); s->d_rows.release();
  s->d_child_ptr.release(); s->d_child_list.release(); s->d_relpos_ptr.release(); s->d_relpos.release();
  s->d_slot_ld.release(); s->d_elim_of_free.release(); s->d_free_of_elim.release(); s->d_level_fronts.release();
  s->d_fail.release(); s->d_sync.release(); s->d_front_off.release(); s->d_slot_dest.release(); s->d_winv_off.release(); s->d_winv.release(); s->d_desc.release(); s->d_cinv.release(); s->d_edge_z.release(); s->d_edge_u.release();
  s->d_edge_lin.release(); s->d_edge_cost.release(); s->d_Hg.release(); s->d_fronts.release(); s->d_fronts_b.release(); s->d_deferred.release(); s->d_pack.release();
  s->d_x.release(); s->d_cand.release(); s->d_scale.release(); s->d_diag.release(); s->d_rhs.release();
  s->d_step.release(); s->d_delta.release(); s->d_scal.release(); s->d_upd.release(); s->d_fsb.release(); s->d_partial.release(); s->d_Hg_alt.release(); s->d_best.release();
  for (auto & row : s->ev_phase) {for (auto & e : row) {if (e) {(void)hipEventDestroy(e);}}}
  for (auto & row : s->ev_lin) {for (auto & e : row) {if (e) {(void)hipEventDestroy(e);}}}
  if (s->h_scal) {(void)hipHostFree(s->h_scal);}
  if (s->h_res) {(void)hipHostFree(s->h_res);}
  if (s->h_upload) {(void)hipHostFree(s->h_upload);}
  if (s->h_fail) {(void)hipHostFree(s->h_fail);}
  if (s->stream2) {(void)hipStreamSynchronize(s->stream2); (void)hipStreamDestroy(s->stream2);}
  for (auto & e : s->ev_level) {if (e) {(void)hipEventDestroy(e);}}
  if (s->stream) {(void)hipStreamDestroy(s->stream);}
  delete s;
}

int kh_spa_set_options(kh_spa * s, const kh_spa_options * o)
{
  if (!s || !o) {return KH_ERR_INVALID_ARG;}
  s->opt = *o;
  return KH_OK;
}

int kh_spa_set_sharding(kh_spa * s, int32_t rank, int32_t world, kh_allreduce_fn fn, void * user)
{
  if (!s || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn)) {return KH_ERR_INVALID_ARG;}
  s->shard_rank = rank; s->shard_world = world; s->allreduce = fn; s->allreduce_user = user; s->comm = nullptr;
  return KH_OK;
}

int kh_spa_set_comm(kh_spa * s, kh_comm * comm)
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  if (comm && kh_comm_device(comm) != s->device) {
    set_error("kh_spa_set_comm: the communicator lives on another device than the solver");
    return KH_ERR_INVALID_ARG;
  }
  s->comm = comm; s->allreduce = nullptr; s->allreduce_user = nullptr;
  s->shard_rank = comm ? kh_comm_rank(comm) : 0;
  s->shard_world = comm ? kh_comm_world(comm) : 1;
  return KH_OK;
}

int kh_spa_set_debug(kh_spa * s, int32_t flags)
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  s->debug_flags = flags;
  return KH_OK;
}

int kh_spa_reset(kh_spa * s)     // ceres_solver.cpp:279-314
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  s->nodes.clear(); s->index_of.clear(); s->cons.clear(); s->con_of.clear(); s->incident.clear(); s->n_dead = 0; s->n_dead_nodes = 0;
  s->corr_ids.clear(); s->corr_poses.clear();
  s->has_first = false; s->was_constant_set = false; s->topology_dirty = true; s->fixed_index = -1;
  s->cached_sn_ids.clear(); s->reuse_count = 0;        // a rebuilt graph is dissected from scratch
  return KH_OK;
}

int kh_spa_clear(kh_spa * s)     // ceres_solver.cpp:272-276
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  s->corr_ids.clear(); s->corr_poses.clear();
  return KH_OK;
}

int kh_spa_add_node(kh_spa * s, int32_t id, const double pose[3])    // ceres_solver.cpp:317-336
{
  if (!s || !pose) {return KH_ERR_INVALID_ARG;}
  if (s->index_of.count(id)) {return KH_OK;}       // unordered_map::insert keeps the existing entry
  Node n; n.id = id; std::copy(pose, pose + 3, n.pose);
  s->index_of[id] = static_cast<int32_t>(s->nodes.size());
  s->nodes.push_back(n);
  // the reference sets first_node_ whenever nodes_->size() == 1 after the insert (ceres_solver.cpp:331-334); tombstoned
  // nodes are already gone from its map, so they do not count here either
  if (static_cast<int32_t>(s->nodes.size()) - s->n_dead_nodes == 1) {s->first_id = id; s->has_first = true;}
  s->topology_dirty = true;
  return KH_OK;
}

static int add_constraint_information(kh_spa * s, int32_t id_a, int32_t id_b, const double * z, const double * omega)
{
  if (!s->index_of.count(id_a) || !s->index_of.count(id_b) || id_a == id_b) {
    set_error("CeresSolver: Failed to add constraint, could not find nodes.");
    return KH_ERR_NOT_FOUND;
  }
  Constraint c; c.a = id_a; c.b = id_b;
  std::copy(z, z + 3, c.z);
  std::copy(omega, omega + 6, c.omega);
  sqrt_information(c.omega, c.u);
  if (!(c.u[0] > 0.0 && c.u[4] > 0.0 && c.u[8] > 0.0)) {     // also false for NaN: llt() of a matrix that is not positive definite
    set_error("CeresSolver: constraint information matrix is not positive definite");
    return KH_ERR_INVALID_ARG;
  }
  s->con_of.insert({kh_spa::edge_key(id_a, id_b), static_cast<int32_t>(s->cons.size())});
  s->incident[id_a].push_back(static_cast<int32_t>(s->cons.size()));
  s->incident[id_b].push_back(static_cast<int32_t>(s->cons.size()));
  s->cons.push_back(c);
  s->topology_dirty = true;
  return KH_OK;
}

int kh_spa_add_constraint(kh_spa * s, int32_t id_a, int32_t id_b, const double z[3], const double cov[9])   // :339-392
{
  if (!s || !z || !cov) {return KH_ERR_INVALID_ARG;}
  double omega[6];
  information_from_covariance(cov, omega);
  return add_constraint_information(s, id_a, id_b, z, omega);
}

int kh_spa_add_constraint_information(kh_spa * s, int32_t id_a, int32_t id_b, const double z[3], const double info_upper[6])
{
  if (!s || !z || !info_upper) {return KH_ERR_INVALID_ARG;}
  return add_constraint_information(s, id_a, id_b, z, info_upper);
}

int kh_spa_get_constraint(kh_spa * s, int32_t index, int32_t * id_a, int32_t * id_b, double z[3], double info_upper[6])
{
  settle(s);
  if (!s || index < 0) {return KH_ERR_INVALID_ARG;}
  if (index >= static_cast<int32_t>(s->cons.size())) {return KH_ERR_NOT_FOUND;}
  const Constraint & c = s->cons[index];
  if (id_a) {*id_a = c.a;}
  if (id_b) {*id_b = c.b;}
  if (z) {std::copy(c.z, c.z + 3, z);}
  if (info_upper) {std::copy(c.omega, c.omega + 6, info_upper);}
  return KH_OK;
}

int kh_spa_get_nodes(kh_spa * s, int32_t * ids, double * poses)
{
  settle(s);
  if (!s) {return KH_ERR_INVALID_ARG;}
  for (size_t k = 0; k < s->nodes.size(); ++k) {
    if (ids) {ids[k] = s->nodes[k].id;}
    if (poses) {std::copy(s->nodes[k].pose, s->nodes[k].pose + 3, poses + 3 * k);}
  }
  return KH_OK;
}

int kh_spa_get_node_at(kh_spa * s, int32_t index, int32_t * id, double pose[3])
{
  settle(s);
  if (!s || index < 0) {return KH_ERR_INVALID_ARG;}
  if (index >= static_cast<int32_t>(s->nodes.size())) {return KH_ERR_NOT_FOUND;}
  if (id) {*id = s->nodes[index].id;}
  if (pose) {std::copy(s->nodes[index].pose, s->nodes[index].pose + 3, pose);}
  return KH_OK;
}

// ---- pose-graph files --------------------------------------------------------------
// The reference persists the graph as a Boost binary archive of the whole Mapper (Mapper.cpp:2635-2651,
// serialization.hpp:38-82) and rebuilds the solver from it with Reset / AddNode* / AddConstraint*
// (slam_toolbox_common.cpp:959-1016).  Boost archives are neither portable nor readable without Boost, so
// the solver's own state is stored instead: g2o SE2 text (VERTEX_SE2 / EDGE_SE2 / FIX), or the same arrays
// as a little-endian binary blob.
namespace
{
const char kBinaryMagic[8] = {'K', 'H', 'P', 'G', 1, 0, 0, 0};

struct GraphFile
{
  std::vector<Node> nodes;
  std::vector<Constraint> cons;      // a, b, z, omega filled
  bool has_fix = false; int32_t fix_id = 0;
};

bool read_all(const char * path, std::string & out)
{
  FILE * f = std::fopen(path, "rb");
  if (!f) {return false;}
  char buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) {out.append(buf, n);}
  const bool ok = !std::ferror(f);
  std::fclose(f);
  return ok;
}

// one whitespace-separated token of [p, end); returns false at end of line
bool next_token(const char *& p, const char * end, const char *& tok, size_t & len)
{
  while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) {++p;}
  if (p >= end) {return false;}
  tok = p;
  while (p < end && *p != ' ' && *p != '\t' && *p != '\r') {++p;}
  len = static_cast<size_t>(p - tok);
  return true;
}

bool parse_numbers(const char *& p, const char * end, int n_int, int32_t * iv, int n_dbl, double * dv)
{
  const char * tok; size_t len; char tmp[64];
  for (int k = 0; k < n_int + n_dbl; ++k) {
    if (!next_token(p, end, tok, len) || len >= sizeof(tmp)) {return false;}
    std::memcpy(tmp, tok, len); tmp[len] = 0;
    char * stop = nullptr;
    if (k < n_int) {
      const long v = std::strtol(tmp, &stop, 10);
      if (stop != tmp + len || v < INT32_MIN || v > INT32_MAX) {return false;}
      iv[k] = static_cast<int32_t>(v);
    } else {
      dv[k - n_int] = std::strtod(tmp, &stop);
      if (stop != tmp + len) {return false;}
    }
  }
  return true;
}

int parse_text(const std::string & data, GraphFile & g)
{
  const char * p = data.data();
  const char * const end_all = p + data.size();
  int line_no = 0;
  char msg[160];
  while (p < end_all) {
    const char * eol = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end_all - p)));
    const char * end = eol ? eol : end_all;
    ++line_no;
    const char * q = p; const char * tok; size_t len;
    p = eol ? eol + 1 : end_all;
    if (!next_token(q, end, tok, len) || tok[0] == '#') {continue;}
    const std::string tag(tok, len);
    bool ok = true;
    if (tag == "VERTEX_SE2") {
      Node n;
      ok = parse_numbers(q, end, 1, &n.id, 3, n.pose);
      if (ok) {g.nodes.push_back(n);}
    } else if (tag == "EDGE_SE2") {
      Constraint c; int32_t ab[2]; double v[9];
      ok = parse_numbers(q, end, 2, ab, 9, v);
      if (ok) {
        c.a = ab[0]; c.b = ab[1];
        std::copy(v, v + 3, c.z); std::copy(v + 3, v + 9, c.omega);
        g.cons.push_back(c);
      }
    } else if (tag == "FIX") {
      int32_t id;
      ok = parse_numbers(q, end, 1, &id, 0, nullptr);
      if (ok && g.has_fix && g.fix_id != id) {
        set_error("kh_spa_load: more than one FIX vertex (the plugin holds exactly the first-added node, ceres_solver.cpp:228-241)");
        return KH_ERR_INVALID_ARG;
      }
      if (ok) {g.has_fix = true; g.fix_id = id;}
    } else {
      std::snprintf(msg, sizeof(msg), "kh_spa_load: line %d: unsupported record '%.40s'", line_no, tag.c_str());
      set_error(msg);
      return KH_ERR_INVALID_ARG;
    }
    if (ok && next_token(q, end, tok, len)) {ok = false;}       // trailing fields
    if (!ok) {
      std::snprintf(msg, sizeof(msg), "kh_spa_load: line %d: malformed %.40s record", line_no, tag.c_str());
      set_error(msg);
      return KH_ERR_INVALID_ARG;
    }
  }
  return KH_OK;
}

int parse_binary(const std::string & data, GraphFile & g)
{
  const size_t header = 8 + 16;
  int64_t n = 0, m = 0;
  if (data.size() >= header) {std::memcpy(&n, data.data() + 8, 8); std::memcpy(&m, data.data() + 16, 8);}
  if (data.size() < header || n < 0 || m < 0 || n > INT32_MAX || m > INT32_MAX ||
    data.size() != header + static_cast<size_t>(n) * (4 + 24) + static_cast<size_t>(m) * (8 + 24 + 48))
  {
    set_error("kh_spa_load: truncated or oversized binary pose-graph file");
    return KH_ERR_INVALID_ARG;
  }
  const char * p = data.data() + header;
  g.nodes.resize(n); g.cons.resize(m);
  for (int64_t i = 0; i < n; ++i) {std::memcpy(&g.nodes[i].id, p + 4 * i, 4);}
  p += 4 * n;
  for (int64_t i = 0; i < n; ++i) {std::memcpy(g.nodes[i].pose, p + 24 * i, 24);}
  p += 24 * n;
  for (int64_t k = 0; k < m; ++k) {std::memcpy(&g.cons[k].a, p + 4 * k, 4);}
  p += 4 * m;
  for (int64_t k = 0; k < m; ++k) {std::memcpy(&g.cons[k].b, p + 4 * k, 4);}
  p += 4 * m;
  for (int64_t k = 0; k < m; ++k) {std::memcpy(g.cons[k].z, p + 24 * k, 24);}
  p += 24 * m;
  for (int64_t k = 0; k < m; ++k) {std::memcpy(g.cons[k].omega, p + 48 * k, 48);}
  return KH_OK;
}
}  // namespace

int kh_spa_save(kh_spa * s, const char * path, int32_t format)
{
  settle(s);
  if (!s || !path || (format != KH_GRAPH_TEXT && format != KH_GRAPH_BINARY)) {return KH_ERR_INVALID_ARG;}
  FILE * f = std::fopen(path, format == KH_GRAPH_BINARY ? "wb" : "w");
  if (!f) {set_error("kh_spa_save: cannot open the file for writing"); return KH_ERR_IO;}
  bool ok = true;
  if (format == KH_GRAPH_TEXT) {
    ok &= std::fprintf(f, "# kartohip pose graph: g2o SE2 records, nodes in AddNode order, the first one is the gauge\n") > 0;
    for (const Node & n : s->nodes) {
      ok &= std::fprintf(f, "VERTEX_SE2 %d %.17g %.17g %.17g\n", n.id, n.pose[0], n.pose[1], n.pose[2]) > 0;
    }
    if (s->has_first && !s->nodes.empty()) {ok &= std::fprintf(f, "FIX %d\n", s->first_id) > 0;}
    for (const Constraint & c : s->cons) {
      ok &= std::fprintf(f, "EDGE_SE2 %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c.a, c.b,
          c.z[0], c.z[1], c.z[2], c.omega[0], c.omega[1], c.omega[2], c.omega[3], c.omega[4], c.omega[5]) > 0;
    }
  } else {
    const int64_t n = static_cast<int64_t>(s->nodes.size()), m = static_cast<int64_t>(s->cons.size());
    std::string blob(kBinaryMagic, 8);
    blob.append(reinterpret_cast<const char *>(&n), 8); blob.append(reinterpret_cast<const char *>(&m), 8);
    for (const Node & nd : s->nodes) {blob.append(reinterpret_cast<const char *>(&nd.id), 4);}
    for (const Node & nd : s->nodes) {blob.append(reinterpret_cast<const char *>(nd.pose), 24);}
    for (const Constraint & c : s->cons) {blob.append(reinterpret_cast<const char *>(&c.a), 4);}
    for (const Constraint & c : s->cons) {blob.append(reinterpret_cast<const char *>(&c.b), 4);}
    for (const Constraint & c : s->cons) {blob.append(reinterpret_cast<const char *>(c.z), 24);}
    for (const Constraint & c : s->cons) {blob.append(reinterpret_cast<const char *>(c.omega), 48);}
    ok &= std::fwrite(blob.data(), 1, blob.size(), f) == blob.size();
  }
  ok &= std::fclose(f) == 0;
  if (!ok) {set_error("kh_spa_save: write failed"); return KH_ERR_IO;}
  return KH_OK;
}

int kh_spa_load(kh_spa * s, const char * path)
{
  if (!s || !path) {return KH_ERR_INVALID_ARG;}
  std::string data;
  if (!read_all(path, data)) {set_error("kh_spa_load: cannot read the file"); return KH_ERR_IO;}
  GraphFile g;
  const bool binary = data.size() >= 8 && std::memcmp(data.data(), kBinaryMagic, 8) == 0;
  const int rc = binary ? parse_binary(data, g) : parse_text(data, g);
  if (rc) {return rc;}
  // validate before touching the solver: a bad file leaves the current graph in place
  std::unordered_map<int32_t, int32_t> seen;
  for (const Node & n : g.nodes) {
    if (!seen.insert({n.id, 0}).second) {set_error("kh_spa_load: duplicate vertex id"); return KH_ERR_INVALID_ARG;}
  }
  for (const Constraint & c : g.cons) {
    if (!seen.count(c.a) || !seen.count(c.b) || c.a == c.b) {
      set_error("kh_spa_load: edge between unknown vertices"); return KH_ERR_INVALID_ARG;
    }
    double u[9];
    sqrt_information(c.omega, u);
    if (!(u[0] > 0.0 && u[4] > 0.0 && u[8] > 0.0)) {     // also false for NaN
      set_error("kh_spa_load: information matrix is not positive definite"); return KH_ERR_INVALID_ARG;
    }
  }
  if (g.has_fix && (g.nodes.empty() || g.fix_id != g.nodes.front().id)) {
    set_error("kh_spa_load: FIX must name the first vertex (the plugin holds the first-added node, ceres_solver.cpp:228-241)");
    return KH_ERR_INVALID_ARG;
  }
  kh_spa_reset(s);                                     // loadSerializedPoseGraph: Reset, AddNode*, AddConstraint*
  for (const Node & n : g.nodes) {kh_spa_add_node(s, n.id, n.pose);}
  for (const Constraint & c : g.cons) {add_constraint_information(s, c.a, c.b, c.z, c.omega);}
  return KH_OK;
}

int kh_spa_remove_node(kh_spa * s, int32_t id)     // ceres_solver.cpp:395-427 (RemoveParameterBlock drops its residuals)
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  auto it = s->index_of.find(id);
  if (it == s->index_of.end()) {set_error("RemoveNode: Failed to find node matching id"); return KH_ERR_NOT_FOUND;}
  auto inc = s->incident.find(id);                     // O(degree): the node's own constraint list
  if (inc != s->incident.end()) {
    for (int32_t k : inc->second) {bury_constraint(s, k);}
    s->incident.erase(inc);
  }
  s->nodes[it->second].dead = 1;
  ++s->n_dead;
  ++s->n_dead_nodes;
  // ceres_solver.cpp:395-427 erases the node and its parameter blocks; first_node_ keeps pointing at the erased entry
  // there (never dereferenced again once the blocks were set constant).  Here the gauge simply ends with the node: no
  // later node takes it over, and the pose-graph files stop naming it.
  if (s->has_first && id == s->first_id) {s->has_first = false;}
  s->index_of.erase(it);
  s->topology_dirty = true;
  return KH_OK;
}

int kh_spa_remove_constraint(kh_spa * s, int32_t id_a, int32_t id_b)    // ceres_solver.cpp:430-448
{
  if (!s) {return KH_ERR_INVALID_ARG;}
  int32_t k = s->first_constraint(id_a, id_b);
  if (k < 0) {k = s->first_constraint(id_b, id_a);}
  if (k < 0) {set_error("RemoveConstraint: Failed to find residual block"); return KH_ERR_NOT_FOUND;}
  bury_constraint(s, k);
  return KH_OK;
}

int kh_spa_modify_node(kh_spa * s, int32_t id, const double pose[3])    // ceres_solver.cpp:451-461
{
  if (!s || !pose) {return KH_ERR_INVALID_ARG;}
  auto it = s->index_of.find(id);
  if (it == s->index_of.end()) {return KH_ERR_NOT_FOUND;}
  Node & n = s->nodes[it->second];
  const double yaw_init = n.pose[2];
  n.pose[0] = pose[0]; n.pose[1] = pose[1]; n.pose[2] = pose[2];
  n.pose[2] += yaw_init;
  return KH_OK;
}

int kh_spa_get_node(kh_spa * s, int32_t id, double pose[3])
{
  if (!s || !pose) {return KH_ERR_INVALID_ARG;}
  auto it = s->index_of.find(id);
  if (it == s->index_of.end()) {return KH_ERR_NOT_FOUND;}
  std::copy(s->nodes[it->second].pose, s->nodes[it->second].pose + 3, pose);
  return KH_OK;
}

int32_t kh_spa_num_nodes(kh_spa * s) {settle(s); return s ? static_cast<int32_t>(s->nodes.size()) : 0;}
int32_t kh_spa_num_constraints(kh_spa * s) {settle(s); return s ? static_cast<int32_t>(s->cons.size()) : 0;}

int kh_spa_iteration_log(kh_spa * s, int32_t capacity, double * rows, int32_t * n_rows)
{
  if (!s || !n_rows || capacity < 0 || (capacity > 0 && !rows)) {return KH_ERR_INVALID_ARG;}
  *n_rows = static_cast<int32_t>(s->iter_log.size());
  const int32_t n = std::min(capacity, *n_rows);
  for (int32_t i = 0; i < n; ++i) {std::copy(s->iter_log[i].begin(), s->iter_log[i].end(), rows + 8 * static_cast<size_t>(i));}
  return KH_OK;
}

int kh_spa_get_corrections(kh_spa * s, int32_t * n, int32_t * ids, double * poses)
{
  if (!s || !n) {return KH_ERR_INVALID_ARG;}
  *n = static_cast<int32_t>(s->corr_ids.size());
  if (ids) {std::copy(s->corr_ids.begin(), s->corr_ids.end(), ids);}
  if (poses) {std::copy(s->corr_poses.begin(), s->corr_poses.end(), poses);}
  return KH_OK;
}

int kh_link_info(const double pose1[3], const double pose2[3], const double cov[9], double diff[3], double cov_out[9])
{
  if (!pose1 || !pose2 || !cov || !diff || !cov_out) {return KH_ERR_INVALID_ARG;}
  // LinkInfo::Update, Mapper.h:174-188; Transform(rPose1, Pose2()), Karto.h:3003-3024
  const double x1 = pose1[0], y1 = pose1[1], t1 = pose1[2];
  double c = 1.0, sn = 0.0, tx = 0.0, ty = 0.0, tth = 0.0;
  if (!(x1 == 0.0 && y1 == 0.0 && t1 == 0.0)) {
    ::sincos(0.0 - t1, &sn, &c);             // one sincos like the reference's GCC build (see ref_sincos in matcher_host.cpp)
    if (x1 != 0.0 || y1 != 0.0) {
      tx = 0.0 - (c * x1 + (0.0 - sn) * y1 + 0.0 * t1);
      ty = 0.0 - (sn * x1 + c * y1 + 0.0 * t1);
    }
    tth = 0.0 - t1;
  }
  diff[0] = tx + (c * pose2[0] + (0.0 - sn) * pose2[1] + 0.0 * pose2[2]);
  diff[1] = ty + (sn * pose2[0] + c * pose2[1] + 0.0 * pose2[2]);
  diff[2] = karto_normalize_angle(pose2[2] + tth);
  // covariance rotated into the frame of pose1: R(-t1) * cov * R(-t1)^T (Matrix3 triple-loop products)
  double cr, sr;
  ::sincos(-t1, &sr, &cr);
  const double omc = 1.0 - cr;      // Matrix3::FromAxisAngle(0, 0, 1, -t1), Karto.h:2482-2511
  const double R[9] = {0.0 * omc + cr, 0.0 - sr, 0.0, 0.0 + sr, 0.0 * omc + cr, 0.0, 0.0, 0.0, 1.0 * omc + cr};
  double tmp[9], Rt[9];
  for (int r = 0; r < 3; ++r) {for (int q = 0; q < 3; ++q) {Rt[3 * r + q] = R[3 * q + r];}}
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {
      tmp[3 * r + q] = R[3 * r] * cov[q] + R[3 * r + 1] * cov[3 + q] + R[3 * r + 2] * cov[6 + q];
    }
  }
  for (int r = 0; r < 3; ++r) {
    for (int q = 0; q < 3; ++q) {
      cov_out[3 * r + q] = tmp[3 * r] * Rt[q] + tmp[3 * r + 1] * Rt[3 + q] + tmp[3 * r + 2] * Rt[6 + q];
    }
  }
  return KH_OK;
}

// CeresSolver::Compute, ceres_solver.cpp:214-269
int kh_spa_compute(kh_spa * s, kh_spa_summary * summary)
{
  settle(s);
  if (!s) {return KH_ERR_INVALID_ARG;}
  kh_spa_summary sum;
  std::memset(&sum, 0, sizeof(sum));
  sum.usable = 1;
  auto finish = [&](int rc) {if (summary) {*summary = sum;} return rc;};
  if (s->nodes.empty()) {
    set_error("CeresSolver: Ceres was called when there are no nodes. This shouldn't happen.");
    return finish(KH_ERR_NOT_FOUND);
  }
  KS_HIP(hipSetDevice(s->device));
  const auto t_begin = std::chrono::steady_clock::now();
  SpaDev dev;
  std::memset(&dev, 0, sizeof(dev));
  bool has_work = false;
  int rc = prepare_problem(s, dev, has_work);
  if (rc) {return finish(rc);}
  auto store_corrections = [&]() {
    s->corr_ids.clear(); s->corr_poses.clear();
    for (const Node & n : s->nodes) {
      s->corr_ids.push_back(n.id);
      s->corr_poses.insert(s->corr_poses.end(), n.pose, n.pose + 3);
    }
  };
  if (!has_work) {          // nothing to optimise: Ceres returns the input, which is "usable"
    store_corrections();
    return finish(KH_OK);
  }
  const kh_spa_options & opt = s->opt;
  const Symbolic & sym = s->sym;
  hipStream_t st = s->stream;
  double * x = s->d_x.p; double * cand = s->d_cand.p;
  double * scal = s->d_scal.p;
  // the fused step kernel writes the candidate of the free nodes only: the gauge node and unused nodes carry their pose
  KS_HIP(hipMemcpyAsync(cand, x, sizeof(double) * 3 * static_cast<size_t>(dev.n_nodes), hipMemcpyDeviceToDevice, s->stream));
  const int n_levels = static_cast<int>(sym.levels.size());
  sum.nnz_factor = sym.nnz_factor;

  auto fetch = [&]() -> int {
    KS_HIP(hipMemcpyAsync(s->h_scal, scal, sizeof(double) * 16, hipMemcpyDeviceToHost, st));
    KS_HIP(hipMemcpyAsync(s->h_fail, s->d_fail.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    KS_HIP(hipStreamSynchronize(st));
    return KH_OK;
  };
  double lin_ms = 0.0, solve_ms = 0.0;
  auto now = []() {return std::chrono::steady_clock::now();};
  auto ms_since = [](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };

  // ---- iteration zero ----
  auto t0 = now();
  if (opt.loss_function != KH_LOSS_NONE && !(opt.loss_scale > 0.0)) {
    set_error("kh_spa: loss_scale must be positive");
    return finish(KH_ERR_INVALID_ARG);
  }
  dev.loss_kind = opt.loss_function; dev.loss_b = opt.loss_scale * opt.loss_scale; dev.loss_a = opt.loss_scale;
  const int32_t e_lo = static_cast<int32_t>(static_cast<int64_t>(dev.n_edges) * s->shard_rank / s->shard_world);
  const int32_t e_hi = static_cast<int32_t>(static_cast<int64_t>(dev.n_edges) * (s->shard_rank + 1) / s->shard_world);
  const int64_t hg_count = static_cast<int64_t>(s->n_slots) * 9 + static_cast<int64_t>(dev.n_free) * 3;
  int n_lin = 0, n_timed = 0;
  // kh_spa_set_debug bit 1: HIP events around the phases of every LM iteration (kh_spa_summary.*_gpu_ms).  Off by default: an event
  // record between two kernels is a 5-6 us bubble on the stream, three of them per iteration were 0.15 ms of a 9 ms solve.
  const bool phase_events = (s->debug_flags & 2) != 0;
  auto allreduce_Hg = [&](const SpaDev & into) -> int {
    if (s->comm) {
      // H || g of this rank's edge block -> sums over all ranks, on the solver's stream (RCCL over xGMI)
      const int arc = kh_comm_allreduce_sum_f64(s->comm, into.H, hg_count, st);
      if (arc) {return arc;}
    } else if (s->shard_world > 1) {
      if (!s->allreduce) {set_error("kh_spa: sharding enabled without a communicator or an all-reduce callback"); return KH_ERR_INVALID_ARG;}
      if (s->allreduce(s->allreduce_user, into.H, hg_count, st) != 0) {set_error("kh_spa: all-reduce callback failed"); return KH_ERR_SOLVER;}
    }
    return KH_OK;
  };
  auto linearize = [&](const SpaDev & into, const double * at, double * cost_slot) -> int {
    const bool timed = phase_events && n_lin < 2 * kh_spa::kMaxTimed + 2;
    if (timed) {KS_HIP(hipEventRecord(s->ev_lin[n_lin][0], st));}
    spa_launch_linearize(into, at, cost_slot, e_lo, e_hi, st);
    if (timed) {KS_HIP(hipEventRecord(s->ev_lin[n_lin][1], st)); ++n_lin;}
    return allreduce_Hg(into);
  };
  // The normal equations at the CANDIDATE point are built speculatively, into a second H || g, in the same batch of
  // launches that evaluates the candidate's cost: a step is nearly always accepted, and the iteration then needs one
  // read-back of scalars instead of two (each was a stream drain plus ~90 us before the next launch reached the GPU).
  // A rejected step simply leaves the second buffer unused.
  SpaDev alt = dev;
  alt.H = s->d_Hg_alt.p; alt.g = s->d_Hg_alt.p + static_cast<size_t>(s->n_slots) * 9;
  // how a front's update matrix reaches its parent (kh_spa_set_debug bits 8, 9): added in by k_syrk (default), read in place by the
  // parent, or summed in by an extend-add launch per level (rounds 3-5)
  const int factor_mode = ((s->debug_flags >> 4) & 15) ? ((s->debug_flags >> 4) & 15) : 3;
  const bool pipeline = factor_mode >= 3 && spa_level_pipeline_fits(sym.max_m, sym.max_ns);
  dev.gather = pipeline && (s->debug_flags & 256) ? 1 : 0;
  dev.scatter = pipeline && !dev.gather && !(s->debug_flags & 512) ? 1 : 0;
  // The fronts (and buffer B) start every factorisation as zeros: 2 x 190 MB on the 10k-node graph, 27 us each at the head of
  // the critical stream.  In scatter mode the fronts are SELF-CLEANING instead: every entry has exactly one last reader in a
  // factorisation (k_potrf the pivot block, k_trsm buffer B under L21, k_syrk the update block, k_backward3 L21), and that reader
  // stores a zero behind itself; only the update matrices that stay in place take a (small) kernel of their own.  (Zeroing on a
  // second stream beside the step evaluation was tried first: the fill kernels take every compute unit, k_step_fused and the
  // linearisation ran 2-3 x longer and the iteration gained 10 us of the 54.)
  auto zero_fronts = [&]() -> int {
    if (dev.scatter && s->clean_a == dev.fronts && s->clean_b == dev.fronts_b) {s->clean_a = nullptr; return KH_OK;}
    s->clean_a = nullptr;
    // (the WHOLE allocation: a later analysis may lay larger fronts over the same buffers, and "clean" has to mean all of them)
    KS_HIP(hipMemsetAsync(dev.fronts, 0, sizeof(double) * (dev.scatter ? s->d_fronts.cap : static_cast<size_t>(dev.fronts_size)), st));
    if (dev.scatter) {KS_HIP(hipMemsetAsync(dev.fronts_b, 0, sizeof(double) * s->d_fronts_b.cap, st));}
    return KH_OK;
  };
  rc = linearize(dev, x, scal + 0); if (rc) {return finish(rc);}
  if (opt.jacobi_scaling) {
    spa_launch_jacobi_scale(dev, s->d_scale.p, st);
  } else {
    std::vector<double> ones(static_cast<size_t>(dev.n_free) * 3, 1.0);
    KS_HIP(hipMemcpyAsync(s->d_scale.p, ones.data(), ones.size() * 8, hipMemcpyHostToDevice, st));
    KS_HIP(hipStreamSynchronize(st));
  }
  spa_launch_grad_norms(dev, x, scal + 1, st);
  KS_HIP(hipMemsetAsync(s->d_fail.p, 0, sizeof(int32_t), st));
  rc = fetch(); if (rc) {return finish(rc);}
  lin_ms += ms_since(t0);
  double x_cost = s->h_scal[0], gmax = s->h_scal[1], x_norm = std::sqrt(s->h_scal[2]);
  sum.initial_cost = x_cost;
  double minimum_cost = x_cost;
  std::vector<double> best_x(static_cast<size_t>(dev.n_nodes) * 3);
  for (int32_t i = 0; i < dev.n_nodes; ++i) {std::copy(s->nodes[i].pose, s->nodes[i].pose + 3, best_x.begin() + 3 * i);}
  bool best_is_current = true;     // device x holds the minimum-cost iterate
  bool best_on_device = false;     // ... otherwise d_best does (best_x on the host until the first accepted step)
  // TrustRegionStepEvaluator
  const int max_nonmono = opt.use_nonmonotonic_steps ? opt.max_consecutive_nonmonotonic_steps : 0;
  double ev_min = x_cost, ev_cur = x_cost, ev_ref = x_cost, ev_cand = x_cost, ev_acc_ref = 0.0, ev_acc_cand = 0.0;
  int ev_nonmono = 0;
  // LevenbergMarquardtStrategy
  double radius = opt.initial_trust_region_radius, decrease_factor = 2.0;
  bool reuse_diagonal = false, have_diagonal = false;
  int num_invalid = 0, iteration = 0;
  bool step_successful = true;
  sum.successful_steps = 1;
  sum.termination = 1;
  if (!(x_cost == x_cost) || std::isinf(x_cost)) {   // initial evaluation failed
    sum.usable = 0; sum.termination = 2;
    set_error("CeresSolver: Ceres could not find a usable solution to optimize.");
    return finish(KH_ERR_SOLVER);
  }

  s->iter_log.clear();
  auto trace = [&](double cost, double cand_cost, double model_change, double radius_used, double radius_next, double step_norm, double verdict) {
    s->iter_log.push_back({static_cast<double>(iteration), cost, cand_cost, model_change, radius_used, radius_next, step_norm, verdict});
  };
  while (true) {
    if (iteration >= opt.max_num_iterations) {sum.termination = 1; break;}
    if (step_successful && gmax <= opt.gradient_tolerance) {sum.termination = 0; break;}
    if (radius < opt.min_trust_region_radius) {sum.termination = 0; break;}
    ++iteration;
    step_successful = false;

    // ---- ComputeTrustRegionStep ----
    auto t1 = now();
    const bool new_diagonal = !reuse_diagonal || !have_diagonal;
    have_diagonal = true;
    const bool timed = phase_events && n_timed < kh_spa::kMaxTimed;
    if (timed) {KS_HIP(hipEventRecord(s->ev_phase[n_timed][0], st));}
    rc = zero_fronts(); if (rc) {return finish(rc);}
    // one launch: the scaled + damped matrix into the fronts (forming the LM diagonal on the way when a new one is due), the
    // right-hand side in elimination order (factorisation and forward solve are one kernel per level) and the fail word's reset
    spa_launch_assemble(dev, s->d_scale.p, s->d_diag.p, 1.0 / radius, new_diagonal, opt.min_lm_diagonal, opt.max_lm_diagonal, s->d_rhs.p, s->d_fail.p, st);
    if (!pipeline) {KS_HIP(hipMemsetAsync(s->d_sync.p, 0, sizeof(int32_t) * 4 * static_cast<size_t>(sym.n_fronts), st));}
    static const int ea_limit = std::getenv("KH_SPA_EXTEND_ADD") ? std::atoi(std::getenv("KH_SPA_EXTEND_ADD")) : 128;
    for (int l = 0; l < n_levels; ++l) {
      const int32_t n_level = s->level_offsets[l + 1] - s->level_offsets[l];
      const int32_t * lf = s->d_level_fronts.p + s->level_offsets[l];
      if (pipeline) {
        // The children's update matrices go into the pivot blocks first (all k_potrf reads); the rest of the extend-add
        // runs on a second stream BESIDE the pivot chains and is waited for before the row solves.
        // (measured on the 10k / 30k graph: 14.7 ms per solve with the split against 13.2 without -- two event records and two
        // stream waits per level on the critical stream cost more than the overlap wins; off unless KH_SPA_EA_OVERLAP=1)
        static const bool overlap_ea = std::getenv("KH_SPA_EA_OVERLAP") && std::atoi(std::getenv("KH_SPA_EA_OVERLAP")) != 0;
        const bool split_ea = l > 0 && !dev.gather && !dev.scatter && overlap_ea && s->stream2;
        if (l > 0 && !dev.gather && !dev.scatter) {
          if (split_ea) {
            KS_HIP(hipEventRecord(s->ev_level[0], st));                    // the level below is complete
            KS_HIP(hipStreamWaitEvent(s->stream2, s->ev_level[0], 0));
            spa_launch_extend_add(dev, lf, n_level, s->level_max_m[l], s->stream2, 2, s->level_max_ns[l]);
            KS_HIP(hipEventRecord(s->ev_level[1], s->stream2));
            spa_launch_extend_add(dev, lf, n_level, s->level_max_m[l], st, 1, s->level_max_ns[l]);
          } else {
            spa_launch_extend_add(dev, lf, n_level, s->level_max_m[l], st);
          }
        }
        spa_launch_potrf_level(dev, s->level_offsets[l], n_level, s->level_max_m[l], s->level_max_ns[l], s->d_fail.p, s->d_rhs.p, s->d_upd.p, st);
        if (split_ea) {KS_HIP(hipStreamWaitEvent(st, s->ev_level[1], 0));}
        // the update of the level: the fronts that fit one workgroup's LDS whole in k_front_update (when there are enough of them), the
        // larger ones -- the head of the level -- in k_trsm / k_syrk
        if (s->level_max_nu[l] == 0) {continue;}                   // (the root: nothing below the pivot block)
        const int32_t split = dev.gather ? n_level : s->level_split[l];
        spa_launch_update_level(dev, s->level_offsets[l], split, s->level_max_m[l], s->level_max_ns[l], s->d_rhs.p, s->d_upd.p, st);
        spa_launch_front_update(dev, s->level_offsets[l] + split, n_level - split, s->level_fused_lds[l], s->d_rhs.p, s->d_upd.p, st);
        continue;
      }
      // narrow levels: the extend-add runs chip-wide in its own launch instead of on each front's single CU
      const bool split = l > 0 && n_level <= ea_limit;
      if (split) {spa_launch_extend_add(dev, lf, n_level, s->level_max_m[l], st);}
      spa_launch_factor_level(dev, lf, n_level, s->level_max_m[l], s->level_max_ns[l], s->d_fail.p,
        s->d_rhs.p, s->d_upd.p, s->d_fsb.p, s->d_sync.p + 4 * s->level_offsets[l], split ? 1 : 0, st);
    }
    if (timed) {KS_HIP(hipEventRecord(s->ev_phase[n_timed][1], st));}
    for (int l = n_levels - 1; l >= 0; --l) {
      const int32_t n_level = s->level_offsets[l + 1] - s->level_offsets[l];
      const int32_t * lf = s->d_level_fronts.p + s->level_offsets[l];
      if (pipeline) {
        spa_launch_backward3_level(dev, s->level_offsets[l], n_level, s->level_max_m[l], s->level_max_ns[l], s->d_rhs.p, st);
      } else {
        spa_launch_backward_level(dev, lf, n_level, s->level_max_m[l], s->d_rhs.p, st);
      }
    }
    if (dev.scatter) {
      spa_launch_zero_update_blocks(dev, s->d_deferred.p, s->n_deferred_fronts, s->deferred_max_m, st);
      s->clean_a = dev.fronts; s->clean_b = dev.fronts_b;
    }
    static const bool speculate = !(std::getenv("KH_SPA_SPECULATE") && std::atoi(std::getenv("KH_SPA_SPECULATE")) == 0);
    static const bool lin_check_env = std::getenv("KH_SPA_CHECK") != nullptr;
    const bool lin_check = lin_check_env || (s->debug_flags & 1);
    bool direct = false;
    const bool fused_step = speculate && !(std::getenv("KH_SPA_FUSED_STEP") && std::atoi(std::getenv("KH_SPA_FUSED_STEP")) == 0);
    if (fused_step) {
      // the candidate's cost AND its normal equations (speculative: a step is nearly always accepted) ride in the same batch
      const bool timed_lin = phase_events && n_lin < 2 * kh_spa::kMaxTimed + 2;
      if (timed_lin) {KS_HIP(hipEventRecord(s->ev_lin[n_lin][0], st));}
      spa_launch_step_and_linearize(dev, alt, s->d_scale.p, s->d_rhs.p, x, s->d_step.p, s->d_delta.p, cand, s->d_partial.p, e_lo, e_hi, st);
      if (timed_lin) {KS_HIP(hipEventRecord(s->ev_lin[n_lin][1], st)); ++n_lin;}
      // a communicator (or a sharding callback) sums this rank's H || g with the others', also with one rank (identity)
      rc = allreduce_Hg(alt); if (rc) {return finish(rc);}
      direct = s->h_res != nullptr && !lin_check;
      if (direct) {s->res_seq = s->res_seq == 0x7fffffff ? 1 : s->res_seq + 1;}
      spa_launch_step_scalars(alt, cand, s->d_partial.p, e_lo > 0 || e_hi < dev.n_edges, scal, st, direct ? s->h_res : nullptr, s->h_res_flag, s->d_fail.p,
                              s->res_seq);
      if (lin_check) {spa_launch_lin_check(dev, s->d_scale.p, s->d_diag.p, 1.0 / radius, s->d_step.p, scal + 12, st);}
      if (timed) {KS_HIP(hipEventRecord(s->ev_phase[n_timed][2], st)); ++n_timed;}
    } else {
      spa_launch_finish_step(dev, s->d_scale.p, s->d_rhs.p, s->d_step.p, s->d_delta.p, st);
      if (lin_check) {spa_launch_lin_check(dev, s->d_scale.p, s->d_diag.p, 1.0 / radius, s->d_step.p, scal + 12, st);}
      spa_launch_model(dev, s->d_scale.p, s->d_step.p, scal + 3, st);
      spa_launch_plus(dev, x, s->d_delta.p, cand, scal + 6, st);
      if (timed) {KS_HIP(hipEventRecord(s->ev_phase[n_timed][2], st)); ++n_timed;}
      if (speculate) {
        rc = linearize(alt, cand, scal + 8); if (rc) {return finish(rc);}      // cost of the candidate + its H, g
        spa_launch_grad_norms(alt, cand, scal + 9, st);
      } else {
        spa_launch_cost(dev, cand, scal + 8, st);
      }
    }
    KS_HIP(hipGetLastError());
    if (direct) {
      // the iteration's last kernel writes its scalars to host-coherent memory and raises the flag behind them; the stream is asked
      // now and then so that a failed launch cannot hang the caller
      volatile int32_t * flag = s->h_res_flag;
      uint64_t spins = 0;
      while (*flag != s->res_seq) {
        _mm_pause();
        if ((++spins & 0x3fff) == 0) {
          const hipError_t e = hipStreamQuery(st);
          if (e == hipSuccess) {
            if (*flag == s->res_seq) {break;}
            set_error("kh_spa: the stream drained without the iteration's result flag"); return finish(KH_ERR_HIP);
          }
          if (e != hipErrorNotReady) {set_error(std::string("kh_spa: ") + hipGetErrorString(e)); return finish(KH_ERR_HIP);}
        }
      }
      for (int q = 3; q <= 10; ++q) {s->h_scal[q] = s->h_res[q];}
      s->h_fail[0] = static_cast<int32_t>(s->h_res[11]);
    } else {
      rc = fetch(); if (rc) {return finish(rc);}
    }
    solve_ms += ms_since(t1);
    if (lin_check) {
      std::fprintf(stderr, "[kh_spa] iteration %d: relative residual of the linear solve %.3e (fail flag %d)\n", iteration,
        std::sqrt(s->h_scal[12] / std::max(s->h_scal[13], 1e-300)), s->h_fail[0]);
      sum.worst_linear_residual = std::max(sum.worst_linear_residual, std::sqrt(s->h_scal[12] / std::max(s->h_scal[13], 1e-300)));
    }
    reuse_diagonal = true;
    const double step_dot_g = s->h_scal[3], step_H_step = s->h_scal[4], nonfinite = s->h_scal[5];
    const double step_norm = std::sqrt(s->h_scal[6]);
    const double cand_norm = std::sqrt(s->h_scal[7]);
    double cand_cost = s->h_scal[8];
    const double model_cost_change = -(step_dot_g + 0.5 * step_H_step);
    bool step_valid = s->h_fail[0] == 0 && nonfinite == 0.0 && std::isfinite(model_cost_change) && model_cost_change > 0.0;
    if (!step_valid) {
      ++num_invalid;
      if (num_invalid >= opt.max_num_consecutive_invalid_steps) {
        trace(x_cost, cand_cost, model_cost_change, radius, radius, step_norm, -1.0);     // (one log row per iteration, this one included)
        sum.termination = 2; sum.usable = 0; break;
      }
      trace(x_cost, cand_cost, model_cost_change, radius, radius / decrease_factor, step_norm, -1.0);
      radius = radius / decrease_factor;      // StepIsInvalid -> StepRejected(0)
      decrease_factor *= 2.0;
      continue;
    }
    num_invalid = 0;
    if (!std::isfinite(cand_cost)) {cand_cost = std::numeric_limits<double>::max();}

    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
      trace(x_cost, cand_cost, model_cost_change, radius, radius, step_norm, 2.0); sum.termination = 0; break;
    }
    const double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= opt.function_tolerance * x_cost) {
      trace(x_cost, cand_cost, model_cost_change, radius, radius, step_norm, 3.0); sum.termination = 0; break;
    }

    double quality;
    if (cand_cost >= std::numeric_limits<double>::max()) {
      quality = std::numeric_limits<double>::lowest();
    } else {
      const double rel = (ev_cur - cand_cost) / model_cost_change;
      const double hist = (ev_ref - cand_cost) / (ev_acc_ref + model_cost_change);
      quality = std::max(rel, hist);
    }
    if (quality > opt.min_relative_decrease) {
      // HandleSuccessfulStep
      if (best_is_current) {   // keep a copy of the best iterate before x moves on (device to device, no drain)
        KS_HIP(hipMemcpyAsync(s->d_best.p, x, best_x.size() * 8, hipMemcpyDeviceToDevice, st));
        best_on_device = true;
        best_is_current = false;
      }
      std::swap(x, cand);
      x_norm = cand_norm;
      if (speculate) {
        std::swap(dev.H, alt.H); std::swap(dev.g, alt.g);          // the speculative linearisation is the current one now
        x_cost = s->h_scal[8]; gmax = s->h_scal[9];
      } else {
        rc = linearize(dev, x, scal + 0); if (rc) {return finish(rc);}
        spa_launch_grad_norms(dev, x, scal + 1, st);
        rc = fetch(); if (rc) {return finish(rc);}
        x_cost = s->h_scal[0]; gmax = s->h_scal[1];
      }
      step_successful = true;
      ++sum.successful_steps;
      const double radius_used = radius;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * quality - 1.0, 3));
      radius = std::min(opt.max_trust_region_radius, radius);
      trace(ev_cur, cand_cost, model_cost_change, radius_used, radius, step_norm, 1.0);
      decrease_factor = 2.0;
      reuse_diagonal = false;
      // TrustRegionStepEvaluator::StepAccepted
      ev_cur = cand_cost;
      ev_acc_cand += model_cost_change;
      ev_acc_ref += model_cost_change;
      if (ev_cur < ev_min) {
        ev_min = ev_cur; ev_nonmono = 0; ev_cand = ev_cur; ev_acc_cand = 0.0;
      } else {
        ++ev_nonmono;
        if (ev_cur > ev_cand) {ev_cand = ev_cur; ev_acc_cand = 0.0;}
      }
      if (ev_nonmono == max_nonmono) {ev_ref = ev_cand; ev_acc_ref = ev_acc_cand;}
      if (x_cost < minimum_cost) {minimum_cost = x_cost; best_is_current = true;}
    } else {
      trace(x_cost, cand_cost, model_cost_change, radius, radius / decrease_factor, step_norm, 0.0);
      radius = radius / decrease_factor;      // StepRejected
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }

  if ((s->debug_flags & 1) && dev.scatter && s->clean_a == dev.fronts && iteration > 0) {
    // the self-cleaning fronts must be all zeros now (kh_spa_set_debug bit 0: every test that checks its linear solves checks this)
    KS_HIP(hipMemsetAsync(s->d_fail.p, 0, sizeof(int32_t), st));
    spa_launch_count_nonzero(dev.fronts, dev.fronts_size, s->d_fail.p, st);
    spa_launch_count_nonzero(dev.fronts_b, dev.fronts_size, s->d_fail.p, st);
    KS_HIP(hipMemcpyAsync(s->h_fail, s->d_fail.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    KS_HIP(hipStreamSynchronize(st));
    if (s->h_fail[0] != 0) {
      s->clean_a = nullptr;
      set_error("kh_spa: " + std::to_string(s->h_fail[0]) + " entries of the self-cleaning fronts were left behind by the last factorisation");
      return finish(KH_ERR_SOLVER);
    }
  }
  sum.iterations = iteration;
  sum.final_cost = minimum_cost;
  sum.linearize_ms = lin_ms; sum.solve_ms = solve_ms;
  sum.factor_flops = sym.factor_flops; sum.factorizations = iteration; sum.levels = n_levels;
  sum.symbolic_ms = s->last_symbolic_ms;
  sum.analysis = s->last_symbolic_ms > 0.0 ? (s->last_analysis_incremental ? 2 : 1) : 0;
  {
    // every recorded event has completed: each iteration ended with a stream synchronisation
    float ms = 0.0f;
    for (int k = 0; k < n_timed; ++k) {
      if (hipEventElapsedTime(&ms, s->ev_phase[k][0], s->ev_phase[k][1]) == hipSuccess) {sum.factor_gpu_ms += ms;}
      if (hipEventElapsedTime(&ms, s->ev_phase[k][1], s->ev_phase[k][2]) == hipSuccess) {sum.backward_gpu_ms += ms;}
    }
    for (int k = 0; k < n_lin; ++k) {
      if (hipEventElapsedTime(&ms, s->ev_lin[k][0], s->ev_lin[k][1]) == hipSuccess) {sum.linearize_gpu_ms += ms;}
    }
  }
  if (!sum.usable) {
    sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    set_error("CeresSolver: Ceres could not find a usable solution to optimize.");
    return finish(KH_ERR_SOLVER);          // ceres_solver.cpp:249-254: state and corrections unchanged
  }
  if (best_is_current || best_on_device) {
    KS_HIP(hipMemcpyAsync(best_x.data(), best_is_current ? x : s->d_best.p, best_x.size() * 8, hipMemcpyDeviceToHost, st));
    KS_HIP(hipStreamSynchronize(st));
  }
  for (int32_t i = 0; i < dev.n_nodes; ++i) {std::copy(best_x.begin() + 3 * i, best_x.begin() + 3 * i + 3, s->nodes[i].pose);}
  store_corrections();
  sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return finish(KH_OK);
}

}  // extern "C"
