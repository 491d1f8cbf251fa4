// Symbolic analysis of the pose-graph normal matrix (hot path B): fill-reducing ordering, supernodes (= fronts of the
// multifrontal factorisation), their row structures, the assembly tree and its level schedule.  Host-only code: no HIP.
//
// Reference: solvers/ceres_solver.cpp:214-269 hands the problem to Ceres' SPARSE_NORMAL_CHOLESKY, whose ordering and
// symbolic factorisation happen inside SuiteSparse/CHOLMOD -- a third-party dependency that is not in the reference tree.
// This is the equivalent stage of the MI355X solver.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace kh
{

struct Symbolic
{
  int32_t n_free = 0, n_fronts = 0;
  // fronts are numbered level by level: level l = the contiguous id range levels[l].front() .. levels[l].back()
  std::vector<int32_t> elim_of_free, free_of_elim, sn_of_elim;      // sn_of_elim: elimination position -> front id
  std::vector<int32_t> rows_ptr, rows, parent, level;
  std::vector<int64_t> front_off;
  std::vector<int32_t> front_m, front_ns, front_first;
  std::vector<int32_t> child_ptr, child_list, relpos_ptr, relpos;
  std::vector<std::vector<int32_t>> levels;
  int64_t fronts_size = 0;
  int64_t nnz_factor = 0;
  int64_t factor_flops = 0;      // sum over the fronts of sum_{j < ns} (m - j)^2
  // gather maps of the level pipeline (a front reads its children's update matrices instead of having them added into it):
  // for child number s of front k, cinv[cinv_ptr[k] + s * (front_m[k] / 3) + i] = index of the child's struct row that sits
  // at position i (node units) of front k, or -1
  std::vector<int32_t> cinv_ptr, cinv;
  // round 6: where a front's update matrix goes (the level pipeline's k_syrk adds it straight into the parent front instead of
  // leaving it for an extend-add pass).  Two destination buffers make the sums independent of the order in which the workgroups
  // of one launch arrive: buffer A (the fronts themselves, holding the assembled entries) takes ONE child per level of the tree,
  // buffer B (zero before every factorisation) takes one more, or two while nothing has been added to it yet (0 + x + y is
  // commutative).  Further children of the same level are `deferred`: their update matrices stay where they are and the parent
  // reads them in place through cinv (they come FIRST in the parent's child list).
  std::vector<uint8_t> scatter_mode;     // per front: 0 deferred (or root), 1 into the parent in buffer A, 2 in buffer B
  std::vector<int32_t> n_deferred;       // per front: children it reads in place
  std::vector<uint8_t> has_b;            // per front: a child adds into buffer B
  // inverse of the pivot block's Cholesky factor, one nsp x nsp block per front (nsp = ns rounded up to 16)
  std::vector<int64_t> winv_off;
  int64_t winv_size = 0;
  int32_t max_m = 0, max_ns = 0;
};

struct SymbolicOptions
{
  int32_t leaf_nodes = 24;          // subsets of at most this many nodes are not dissected further (12 / 16 / 24 / 32 on the 10k-node
                                    // graph: 13 / 13 / 12 / 12 levels, factorisations of a solve 9.54 / 9.44 / 8.87 / 8.94 ms)
  int32_t max_pivot_nodes = 42;     // supernodes with more pivots are split into a chain of fronts (42 nodes = 126 columns:
                                    // the pivot block of a front is factored inside one workgroup's LDS, 128 x 130 doubles)
  int32_t separator_candidates = 2; // BFS levels tried as the cut of a subset (each one refined to a minimum vertex cover)
  // parallel loop the nested dissection runs the independent subsets of a tree level on (the library passes its persistent host
  // pool); empty = serial
  std::function<void(size_t, const std::function<void(size_t)> &)> parallel_for;
  double balance_lo = 0.35;         // a cut must leave at least this fraction of the subset on the near side ...
  double balance_hi = 0.65;         // ... and at most this
};

// adjacency of the free nodes in CSR form: neighbours of i = adj_idx[adj_ptr[i] .. adj_ptr[i + 1]), ascending, no
// self loops.  Returns 0 or a KH_ERR_* code (message through kh::set_error).
int build_symbolic(Symbolic & sym, int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt);
// The two halves of build_symbolic.  nested_dissection: the supernodes in elimination order (before the chain split).
// build_structure: everything else, for ANY partition of the free nodes into supernodes in a valid elimination order --
// e.g. the supernodes of an earlier analysis with the nodes that have left removed and the new nodes as leading leaves
// (the solver's incremental re-analysis after a loop closure); `supernodes` is consumed.
int nested_dissection(int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt, std::vector<std::vector<int32_t>> & supernodes);
int build_structure(Symbolic & sym, int32_t n_free, const std::vector<int32_t> & adj_ptr, const std::vector<int32_t> & adj_idx,
  const SymbolicOptions & opt, std::vector<std::vector<int32_t>> & supernodes);

}  // namespace kh
