// Internal structures shared by spa_host.cpp and spa_kernels.hip (pose-graph SPA solver, hot path B).
#pragma once
#include <cstddef>
#include <cstdint>

namespace kh
{

// Everything a workgroup of the level pipeline needs to know about its front, in one 128-byte record (fronts are numbered
// level by level, so a level's records are contiguous), including where its first three children's update matrices are: a
// front READS its children's contributions when it loads its own entries (no extend-add pass over the parent).
struct ChildInfo
{
  int64_t off;             // offset (doubles) of the child front in `fronts`
  int32_t m, ns;           // its dimension and pivot count: the update matrix is the trailing (m - ns)^2 block
  int32_t rows_ptr;        // 3 * rows_ptr = the child's slot in upd (forward-solve contributions)
  int32_t pad;
};
struct alignas(16) FrontDesc
{
  int64_t off;             // offset (doubles) of the front in `fronts`
  int64_t woff;            // offset (doubles) of its L11^-T block in `winv`
  int32_t m, ns;           // scalar dimension, scalar pivot columns
  int32_t first;           // first elimination position (node units)
  int32_t rows_ptr;        // struct rows in front_rows; also 3 * rows_ptr = its slot in upd
  int32_t child_ptr, child_end;      // children in child_list
  int32_t relpos_ptr;      // positions of its struct rows inside the parent front (node units) in relpos
  int32_t parent;
  int32_t cinv_ptr;        // gather maps in cinv: child number s -> cinv[cinv_ptr + s * (m / 3) + position] = child row or -1
  int32_t flags;           // bits 0-1: where its update matrix goes (Symbolic::scatter_mode); bit 2: a child adds into buffer B;
                           // bits 8..: children it reads in place (the first ones of its child list)
  ChildInfo ch[3];
};
static_assert(sizeof(FrontDesc) == 128, "FrontDesc is one 128-byte record");

// Device view of one linear-algebra problem instance.  All pointers are device pointers.
struct SpaDev
{
  // graph
  int32_t n_nodes, n_free, n_edges;
  const int32_t * edge_a;        // node index of the source pose
  const int32_t * edge_b;
  const double * edge_z;         // 3 per edge: measurement (LinkInfo::GetPoseDifference)
  const double * edge_u;         // 9 per edge: upper sqrt-information U, row-major (ceres_solver.cpp:376)
  const int32_t * free_of_node;  // node -> free index or -1 (gauge node / unused node)
  const int32_t * node_of_free;  // free index -> node
  // per-edge linearisation scratch: r(3) Ja(9) Jb(9), row-major
  double * edge_lin;
  double * edge_cost;
  // robust loss rho(s) on s = |U r|^2 (ceres_solver.cpp:82-94): 0 none, 1 Huber(a), 2 Cauchy(a); b = a^2
  int32_t loss_kind;
  double loss_a, loss_b;
  // BSR normal matrix over the free nodes (full symmetric pattern), 9 doubles per block, row-major blocks
  int32_t n_slots;
  const int32_t * slot_contrib_ptr;   // n_slots+1
  const int32_t * slot_contrib;       // edge*4 + kind (0 = Ja^T Ja, 1 = Jb^T Jb, 2 = Ja^T Jb, 3 = Jb^T Ja)
  const int32_t * bsr_row_ptr;        // n_free+1
  const int32_t * bsr_col;            // n_slots
  const int32_t * bsr_diag_slot;      // n_free
  double * H;                         // n_slots*9
  // gradient gather: per free node the incident edges (edge*2 + role)
  const int32_t * node_contrib_ptr;   // n_free+1
  const int32_t * node_contrib;
  double * g;                         // 3*n_free
  // multifrontal structure
  int32_t n_fronts;
  const int64_t * front_off;          // offset (doubles) of front k in `fronts`
  const int32_t * front_m;            // scalar dimension of the front
  const int32_t * front_ns;           // scalar number of pivot columns
  const int32_t * front_first;        // first elimination position (node units) of the supernode
  const int32_t * front_rows_ptr;     // n_fronts+1 : struct rows (node units, elimination positions)
  const int32_t * front_rows;
  const int32_t * child_ptr;          // n_fronts+1
  const int32_t * child_list;
  const int32_t * relpos_ptr;         // per front: positions (node units) of its struct rows inside the parent front
  const int32_t * relpos;
  const int64_t * slot_dest;          // per BSR slot: offset of block (0,0) inside `fronts`, or -1 (upper part)
  const int32_t * slot_ld;            // leading dimension (front m) at that destination
  const int32_t * elim_of_free;       // free index -> elimination position
  const int32_t * free_of_elim;
  double * fronts;
  int64_t fronts_size;
  // round 3: L11^-T of every front (nsp x nsp, nsp = ns rounded up to 16; row q holds (L11^-T)[q][:], zeros left of the diagonal)
  double * winv;
  const int64_t * winv_off;
  const FrontDesc * desc;
  const int32_t * cinv;
  int32_t gather;          // 1: fronts read ALL their children's update matrices in place; 0: they have been summed in
  // round 6: k_syrk adds a front's update matrix into its parent (buffer A = fronts, or buffer B = fronts_b, same offsets; see
  // Symbolic::scatter_mode), no extend-add pass; 0: spa_launch_extend_add sums the update matrices into the fronts
  int32_t scatter;
  double * fronts_b;
};

// [e_lo, e_hi): edge block linearised by this rank (0, n_edges on a single GPU)
void spa_launch_linearize(const SpaDev & d, const double * x, double * cost_out, int e_lo, int e_hi, void * stream);
void spa_launch_cost(const SpaDev & d, const double * x, double * cost_out, void * stream);
void spa_launch_jacobi_scale(const SpaDev & d, double * scale_out, void * stream);
// the head of a factorisation: scaled + damped H into the fronts; with compute_diag the LM diagonal clamp(diag(S H S)) is formed
// (and left in `diagonal`) on the way; rhs (elimination order) <- scale * g; *fail_flag <- 0
void spa_launch_assemble(const SpaDev & d, const double * scale, double * diagonal, double inv_radius, bool compute_diag, double min_diag, double max_diag,
                         double * rhs, int32_t * fail_flag, void * stream);
// max_m = largest front dimension of the level (sizes the LDS panel / vector)
// also does the forward solve of the level: rhs (elimination order) in, y out; upd as for the backward level
// sync: 4 ints per front of the level (zeroed before every factorisation): hand-offs between the workgroups sharing a front
// max_ns = most pivot columns of a front of the level; fsb: 3 * (free nodes + front_rows_ptr[n_fronts]) doubles, where the
// right-hand-side slice of a front waits between the launches of the split form
void spa_launch_factor_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, int32_t max_ns, int32_t * fail_flag,
                             double * rhs, double * upd, double * fsb, int32_t * sync, int32_t matrix_added, void * stream);
// the children's update matrices of every front of the level summed into the fronts, one workgroup per (front, 16
// destination columns); follow with spa_launch_factor_level(..., matrix_added = 1, ...)
// part 0: everything; 1: only the pivot blocks (max_ns = most pivot columns of a front of the level); 2: everything else
void spa_launch_extend_add(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, void * stream, int32_t part = 0,
                           int32_t max_ns = 0);
// upd: per-front forward-solve contributions to the ancestors, 3 * front_rows_ptr[n_fronts] doubles
void spa_launch_backward_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, double * rhs, void * stream);
// round-3 level pipeline (potrf -> trsm -> syrk, see spa_kernels.hip): the children's update matrices must already be summed
// into the fronts (spa_launch_extend_add); also does the forward solve of the level like spa_launch_factor_level
// (the level = fronts first_front .. first_front + n - 1)
void spa_launch_potrf_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, int32_t * fail_flag,
                            double * rhs, double * upd, void * stream);
void spa_launch_update_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, double * rhs, double * upd, void * stream);
// round 6: trsm + syrk of one front per workgroup with L21 resident in LDS (k_front_update), for the fronts first_front .. + n - 1,
// all of which must fit: spa_front_update_lds(m, ns) = the LDS such a front takes, 0 when it does not fit or has nothing to update;
// lds_bytes = the largest of the range
size_t spa_front_update_lds(int32_t m, int32_t ns);
void spa_launch_front_update(const SpaDev & d, int32_t first_front, int32_t n, size_t lds_bytes, const double * rhs, double * upd, void * stream);
void spa_launch_backward3_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, double * rhs, void * stream);
// self-cleaning fronts (scatter mode): zero the update matrices of the fronts in `list` (children read in place by their parents)
void spa_launch_zero_update_blocks(const SpaDev & d, const int32_t * list, int32_t n, int32_t max_m, void * stream);
// debugging aid: *count += entries of p[0..n) whose bit pattern is not zero
void spa_launch_count_nonzero(const double * p, int64_t n, int32_t * count, void * stream);
// whether the largest front of a problem fits the LDS budgets of the level pipeline (otherwise: panel-pair kernels)
bool spa_level_pipeline_fits(int32_t max_m, int32_t max_ns);
// step = -y (free order), delta = step * scale
void spa_launch_finish_step(const SpaDev & d, const double * scale, const double * rhs, double * step, double * delta, void * stream);
// out[0] = step.gs, out[1] = step^T Hs step, out[2] = any non-finite in step
void spa_launch_model(const SpaDev & d, const double * scale, const double * step, double * out3, void * stream);
// step = -y, delta, cand = Plus(x, delta), the model-cost terms and step norms from `cur`; cost and normal equations of
// the candidate (this rank's edge block [e_lo, e_hi)) into `alt`.  A sharded caller sums alt.H || alt.g over the ranks, then
// spa_launch_step_scalars leaves scal[3..7] (step), scal[8] (cost), scal[9..10] (gradient norms).
// partial: spa_step_partials_size doubles.
int64_t spa_step_partials_size(const SpaDev & d);
void spa_launch_step_and_linearize(const SpaDev & cur, const SpaDev & alt, const double * scale, const double * rhs, const double * x, double * step,
                                   double * delta, double * cand, double * partial, int e_lo, int e_hi, void * stream);
// h_out (host-coherent, 16 doubles; nullptr: none): scal[3..10] and the fail word (h_out[11]) also go there, followed by the
// system-scope store h_flag <- seq
void spa_launch_step_scalars(const SpaDev & alt, const double * cand, double * partial, bool sharded, double * scal, void * stream,
                             double * h_out = nullptr, int32_t * h_flag = nullptr, const int32_t * fail_flag = nullptr, int32_t seq = 0);
// debugging aid (KH_SPA_CHECK): out[0] = |(Hs + D / radius) step + gs|^2, out[1] = |gs|^2 from the BSR matrix
void spa_launch_lin_check(const SpaDev & d, const double * scale, const double * diagonal, double inv_radius, const double * step, double * out2,
                          void * stream);
// cand = Plus(x, delta); out[0] = |x - cand|^2 over free params, out[1] = |cand|^2 over free params
void spa_launch_plus(const SpaDev & d, const double * x, const double * delta, double * cand, double * out2, void * stream);
// projected-gradient norms: out[0] = max |x - Plus(x, -g)|, out[1] = |x_free|^2
void spa_launch_grad_norms(const SpaDev & d, const double * x, double * out2, void * stream);

}  // namespace kh
