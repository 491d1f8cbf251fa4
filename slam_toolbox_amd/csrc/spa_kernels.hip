// gfx950 kernels of the pose-graph SPA solver (hot path B).
//
//   k_edge_lin      K5a  PoseGraph2dErrorTerm residual + analytic 3x3 Jacobian blocks per edge
//                        (solvers/ceres_utils.h:84-100; autodiff in the reference)
//   k_gather_H/g    K5b  J^T J / J^T r reduced into the BSR normal matrix and gradient.  Gather form:
//                        one thread per output element walks its (precomputed) contribution list in
//                        a fixed order -> no atomics, bit-reproducible H and g.
//   k_assemble      K6a  scaled + damped H scattered into the multifrontal front storage
//   k_factor        K6b  one workgroup per front of an elimination-tree level: extend-add of the
//                        children's Schur complements, blocked right-looking partial Cholesky
//   k_forward/back  K6c  level-scheduled triangular solves
//
// Numerics here are FP64 with FMA contraction allowed (parity bar for the solver is 1e-9 against the
// CPU restatement, not bit equality).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include "lds_attr.hpp"
#include "spa_internal.hpp"

#pragma clang fp contract(fast)

namespace kh
{

constexpr double kPiD = 3.14159265358979323846;
constexpr double kTwoPiD = 2.0 * kPiD;

// ceres_utils.h:27-32
__device__ __forceinline__ double d_normalize_angle(double a) {return a - kTwoPiD * floor((a + kPiD) / kTwoPiD);}

// ---------------------------------------------------------------------------------------------
// [e_lo, e_hi) = the edge block this GPU linearises (all edges on one GPU); edges outside it get a zero
// record, so the gather kernels below produce this rank's PARTIAL H and g, summed across ranks by the
// caller's all-reduce.
template <int N>
__device__ __forceinline__ void block_reduce_store(double (&v)[N], int max_mask, double * out)      // bit q of max_mask: entry q is a maximum
{
  __shared__ double red[N][256];
#pragma unroll
  for (int q = 0; q < N; ++q) {red[q][threadIdx.x] = v[q];}
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const double a = red[q][threadIdx.x], b = red[q][threadIdx.x + w];
        red[q][threadIdx.x] = ((max_mask >> q) & 1) ? fmax(a, b) : a + b;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < N) {out[threadIdx.x] = red[threadIdx.x][0];}
}

// device half of k_edge_lin: returns rho(s) of the edge (0 for the threads behind the last edge)
template <bool kJac>
__device__ __forceinline__ double edge_lin(const SpaDev & d, const double * __restrict__ x, int e, int e_lo, int e_hi)
{
  if (e >= d.n_edges) {return 0.0;}
  if (kJac && (e < e_lo || e >= e_hi)) {
    double * out = d.edge_lin + 21 * (size_t)e;
#pragma unroll
    for (int q = 0; q < 21; ++q) {out[q] = 0.0;}
    return 0.0;
  }
  const int a = d.edge_a[e], b = d.edge_b[e];
  const double xa = x[3 * a], ya = x[3 * a + 1], ta = x[3 * a + 2];
  const double xb = x[3 * b], yb = x[3 * b + 1], tb = x[3 * b + 2];
  const double c = cos(ta), s = sin(ta);
  const double dx = xb - xa, dy = yb - ya;
  const double * z = d.edge_z + 3 * e;
  const double * U = d.edge_u + 9 * e;
  const double r0 = c * dx + s * dy - z[0];
  const double r1 = -s * dx + c * dy - z[1];
  const double r2 = d_normalize_angle((tb - ta) - z[2]);
  // f = U r (U upper triangular)
  double f0 = U[0] * r0 + U[1] * r1 + U[2] * r2;
  double f1 = U[4] * r1 + U[5] * r2;
  double f2 = U[8] * r2;
  const double sq = f0 * f0 + f1 * f1 + f2 * f2;
  // Robust loss, as Ceres' ResidualBlock::Evaluate + Corrector apply it: cost 0.5 rho(s); both losses the
  // plugin offers have rho'' <= 0, for which the corrector degenerates to scaling the residual and the
  // Jacobian rows by sqrt(rho'(s)).
  double rho0 = sq, w = 1.0;
  if (d.loss_kind == 1) {            // ceres::HuberLoss: s <= b -> (s, 1); else (2 a sqrt(s) - b, a / sqrt(s))
    if (sq > d.loss_b) {
      const double r = sqrt(sq);
      rho0 = 2.0 * d.loss_a * r - d.loss_b;
      w = sqrt(fmax(2.2250738585072014e-308, d.loss_a / r));
    }
  } else if (d.loss_kind == 2) {     // ceres::CauchyLoss: b log(1 + s/b), 1 / (1 + s/b)
    const double sum = 1.0 + sq * (1.0 / d.loss_b);
    rho0 = d.loss_b * log(sum);
    w = sqrt(fmax(2.2250738585072014e-308, 1.0 / sum));
  }
  d.edge_cost[e] = rho0;
  if (!kJac) {return rho0;}
  double * out = d.edge_lin + 21 * (size_t)e;
  f0 *= w; f1 *= w; f2 *= w;
  out[0] = f0; out[1] = f1; out[2] = f2;
  // raw Jacobians (rows = residual, cols = xa, ya, ta | xb, yb, tb)
  const double ja[9] = {-c, -s, -s * dx + c * dy, s, -c, -c * dx - s * dy, 0.0, 0.0, -1.0};
  const double jb[9] = {c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    out[3 + 0 + col] = w * (U[0] * ja[col] + U[1] * ja[3 + col] + U[2] * ja[6 + col]);
    out[3 + 3 + col] = w * (U[4] * ja[3 + col] + U[5] * ja[6 + col]);
    out[3 + 6 + col] = w * (U[8] * ja[6 + col]);
    out[12 + 0 + col] = w * (U[0] * jb[col] + U[1] * jb[3 + col] + U[2] * jb[6 + col]);
    out[12 + 3 + col] = w * (U[4] * jb[3 + col] + U[5] * jb[6 + col]);
    out[12 + 6 + col] = w * (U[8] * jb[6 + col]);
  }
  return rho0;
}

// cost_partial: per-workgroup sums of rho (the step evaluation's form; nullptr: none).  A sharded linearisation (edges outside
// [e_lo, e_hi) get a zero record) does not produce them: the cost is evaluated over all edges by the <false> instance.
template <bool kJac>
__global__ __launch_bounds__(256) void k_edge_lin(SpaDev d, const double * __restrict__ x, int e_lo, int e_hi, double * cost_partial)
{
  double p[1] = {edge_lin<kJac>(d, x, blockIdx.x * blockDim.x + threadIdx.x, e_lo, e_hi)};
  if (cost_partial) {block_reduce_store<1>(p, 0, cost_partial + blockIdx.x);}
}

__device__ __forceinline__ void gather_H(const SpaDev & d, int t)
{
  if (t >= d.n_slots * 9) {return;}
  const int slot = t / 9, el = t - slot * 9;
  const int row = el / 3, col = el - row * 3;
  double acc = 0.0;
  for (int k = d.slot_contrib_ptr[slot]; k < d.slot_contrib_ptr[slot + 1]; ++k) {
    const int code = d.slot_contrib[k];
    const int e = code >> 2, kind = code & 3;
    const double * lin = d.edge_lin + 21 * (size_t)e;
    const double * L = lin + ((kind == 0 || kind == 2) ? 3 : 12);   // left factor (transposed): Ja or Jb
    const double * R = lin + ((kind == 0 || kind == 3) ? 3 : 12);   // right factor
    acc += L[row] * R[col] + L[3 + row] * R[3 + col] + L[6 + row] * R[6 + col];
  }
  d.H[t] = acc;
}
__global__ __launch_bounds__(256) void k_gather_H(SpaDev d) {gather_H(d, blockIdx.x * blockDim.x + threadIdx.x);}

__global__ __launch_bounds__(256) void k_gather_g(SpaDev d)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_free * 3) {return;}
  const int i = t / 3, comp = t - i * 3;
  double acc = 0.0;
  for (int k = d.node_contrib_ptr[i]; k < d.node_contrib_ptr[i + 1]; ++k) {
    const int code = d.node_contrib[k];
    const int e = code >> 1, role = code & 1;
    const double * lin = d.edge_lin + 21 * (size_t)e;
    const double * J = lin + (role ? 12 : 3);
    acc += J[comp] * lin[0] + J[3 + comp] * lin[1] + J[6 + comp] * lin[2];
  }
  d.g[t] = acc;
}

// deterministic single-workgroup sum: out = factor * sum(in[0..n))
__global__ __launch_bounds__(1024) void k_sum(const double * __restrict__ in, int n, double factor, double * out)
{
  __shared__ double s[1024];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {acc += in[i];}
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {s[threadIdx.x] += s[threadIdx.x + w];}
    __syncthreads();
  }
  if (threadIdx.x == 0) {out[0] = factor * s[0];}
}

void spa_launch_linearize(const SpaDev & d, const double * x, double * cost_out, int e_lo, int e_hi, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  if (d.n_edges > 0) {
    if (e_lo > 0 || e_hi < d.n_edges) {
      // sharded: the cost is still evaluated over all edges on every rank (E threads, microseconds)
      hipLaunchKernelGGL(k_edge_lin<false>, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d, x, 0, d.n_edges, (double *)nullptr);
    }
    hipLaunchKernelGGL(k_edge_lin<true>, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d, x, e_lo, e_hi, (double *)nullptr);
    hipLaunchKernelGGL(k_gather_H, dim3((d.n_slots * 9 + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_gather_g, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, s, d);
  }
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, d.edge_cost, d.n_edges, 0.5, cost_out);
}

void spa_launch_cost(const SpaDev & d, const double * x, double * cost_out, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  if (d.n_edges > 0) {
    hipLaunchKernelGGL(k_edge_lin<false>, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d, x, 0, d.n_edges, (double *)nullptr);
  }
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, d.edge_cost, d.n_edges, 0.5, cost_out);
}

// ---------------------------------------------------------------------------------------------
// small vector kernels
__global__ void k_jacobi_scale(SpaDev d, double * scale)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_free * 3) {return;}
  const int i = t / 3, c = t - i * 3;
  const double hii = d.H[(size_t)d.bsr_diag_slot[i] * 9 + c * 3 + c];
  scale[t] = 1.0 / (1.0 + sqrt(hii));      // trust_region_minimizer.cc: jacobian_scaling = 1 / (1 + sqrt(colnorm^2))
}
void spa_launch_jacobi_scale(const SpaDev & d, double * scale_out, void * stream)
{
  hipLaunchKernelGGL(k_jacobi_scale, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, scale_out);
}

// The head of a factorisation in ONE launch (round 6; three launches before): the scaled + damped H into the fronts, the LM
// diagonal itself where a new one is due (compute_diag: clamp(diag(S H S)) to [lo, hi], levenberg_marquardt_strategy.cc -- the entry
// that needs it is the entry that computes it, and it is kept in `diagonal` for the iterations that reuse it), and behind the blocks
// of the matrix the right-hand side rhs (elimination order) <- scale * g and the fail word <- 0.
__global__ __launch_bounds__(256) void k_assemble(SpaDev d, const int32_t * slot_row, const double * scale, double * diagonal, double inv_radius,
                                                  int compute_diag, double lo, double hi, double * rhs, int32_t * fail_flag, int nb_matrix)
{
  if ((int)blockIdx.x >= nb_matrix) {
    const int t = ((int)blockIdx.x - nb_matrix) * 256 + (int)threadIdx.x;
    if (t == 0 && fail_flag) {*fail_flag = 0;}
    if (t >= d.n_free * 3) {return;}
    const int i = t / 3, c = t - i * 3;
    rhs[3 * d.elim_of_free[i] + c] = scale[t] * d.g[t];
    return;
  }
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= d.n_slots * 9) {return;}
  const int slot = t / 9, el = t - slot * 9;
  const int64_t dest = d.slot_dest[slot];
  if (dest < 0) {return;}
  const int r = el / 3, c = el - r * 3;
  const int i = slot_row[slot], j = d.bsr_col[slot];
  // self-cleaning fronts: nothing is written that no kernel of the factorisation reads (and zeroes) -- the entries of a diagonal
  // block above the diagonal would outlive a change of the fronts' layout
  if (d.scatter && i == j && r < c) {return;}
  double v = scale[3 * i + r] * d.H[t] * scale[3 * j + c];
  if (i == j && r == c) {
    double dg;
    if (compute_diag) {
      dg = v < lo ? lo : v;
      dg = dg > hi ? hi : dg;
      diagonal[3 * i + r] = dg;
    } else {
      dg = diagonal[3 * i + r];
    }
    v += dg * inv_radius;
  }
  d.fronts[dest + r + (int64_t)c * d.slot_ld[slot]] = v;
}

// slot_row is stored right behind bsr_col by the host (bsr_col + n_slots)
void spa_launch_assemble(const SpaDev & d, const double * scale, double * diagonal, double inv_radius, bool compute_diag, double min_diag, double max_diag,
                         double * rhs, int32_t * fail_flag, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  // (the fronts have been zeroed by the caller, or clean themselves)
  const int nbm = (d.n_slots * 9 + 255) / 256, nbr = (d.n_free * 3 + 255) / 256;
  hipLaunchKernelGGL(k_assemble, dim3(nbm + nbr), dim3(256), 0, s, d, d.bsr_col + d.n_slots, scale, diagonal, inv_radius, compute_diag ? 1 : 0, min_diag,
                     max_diag, rhs, fail_flag, nbm);
}

__global__ void k_finish_step(SpaDev d, const double * scale, const double * rhs, double * step, double * delta)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_free * 3) {return;}
  const int i = t / 3, c = t - i * 3;
  const double y = rhs[3 * d.elim_of_free[i] + c];
  step[t] = -y;                   // levenberg_marquardt_strategy.cc: step *= -1
  delta[t] = -y * scale[t];       // trust_region_minimizer.cc: delta = step .* jacobian_scaling
}
void spa_launch_finish_step(const SpaDev & d, const double * scale, const double * rhs, double * step, double * delta, void * stream)
{
  hipLaunchKernelGGL(k_finish_step, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, scale, rhs, step, delta);
}

// model cost change pieces: tmp[row] = step_row * gs_row, tmp2[row] = step_row * (Hs step)_row
__global__ void k_model(SpaDev d, const double * scale, const double * step, double * tmp)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n3 = d.n_free * 3;
  if (t >= n3) {return;}
  const int i = t / 3, r = t - i * 3;
  double acc = 0.0;
  for (int k = d.bsr_row_ptr[i]; k < d.bsr_row_ptr[i + 1]; ++k) {
    const int j = d.bsr_col[k];
    const double * blk = d.H + (size_t)k * 9 + r * 3;
    acc += blk[0] * scale[3 * j] * step[3 * j] + blk[1] * scale[3 * j + 1] * step[3 * j + 1] + blk[2] * scale[3 * j + 2] * step[3 * j + 2];
  }
  acc *= scale[t];
  tmp[t] = step[t] * scale[t] * d.g[t];
  tmp[n3 + t] = step[t] * acc;
  tmp[2 * n3 + t] = (step[t] - step[t] == 0.0) ? 0.0 : 1.0;    // non-finite marker
}
void spa_launch_model(const SpaDev & d, const double * scale, const double * step, double * out3, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  const int n3 = d.n_free * 3;
  double * tmp = d.edge_lin;     // scratch: 21*E >= 9*n_free is guaranteed by the host
  hipLaunchKernelGGL(k_model, dim3((n3 + 255) / 256), dim3(256), 0, s, d, scale, step, tmp);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, tmp, n3, 1.0, out3);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, tmp + n3, n3, 1.0, out3 + 1);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, tmp + 2 * n3, n3, 1.0, out3 + 2);
}

__global__ void k_plus(SpaDev d, const double * x, const double * delta, double * cand, double * tmp)
{
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= d.n_nodes) {return;}
  const int f = d.free_of_node[n];
  double px = x[3 * n], py = x[3 * n + 1], pt = x[3 * n + 2];
  if (f >= 0) {
    const double nx = px + delta[3 * f], ny = py + delta[3 * f + 1];
    const double nt = d_normalize_angle(pt + delta[3 * f + 2]);     // AngleLocalParameterization, ceres_utils.h:38-55
    const double ex = px - nx, ey = py - ny, et = pt - nt;
    tmp[f] = ex * ex + ey * ey + et * et;
    tmp[d.n_free + f] = nx * nx + ny * ny + nt * nt;
    px = nx; py = ny; pt = nt;
  }
  cand[3 * n] = px; cand[3 * n + 1] = py; cand[3 * n + 2] = pt;
}
// KH_SPA_CHECK=1 (debugging aid): the linear system of this iteration, (Hs + D / radius) step + gs = 0 in the scaled
// variables, evaluated with the BSR matrix -- independent of the fronts.  out[0] = |residual|^2, out[1] = |gs|^2.
__global__ __launch_bounds__(1024) void k_lin_check(SpaDev d, const double * scale, const double * diagonal, double inv_radius,
                                                     const double * step, double * out2)
{
  __shared__ double s0[1024];
  __shared__ double s1[1024];
  double a0 = 0.0, a1 = 0.0;
  const int n3 = d.n_free * 3;
  for (int t = threadIdx.x; t < n3; t += 1024) {
    const int i = t / 3, r = t - i * 3;
    double acc = 0.0;
    for (int k = d.bsr_row_ptr[i]; k < d.bsr_row_ptr[i + 1]; ++k) {
      const int j = d.bsr_col[k];
      const double * blk = d.H + (size_t)k * 9 + r * 3;
      acc += blk[0] * scale[3 * j] * step[3 * j] + blk[1] * scale[3 * j + 1] * step[3 * j + 1] + blk[2] * scale[3 * j + 2] * step[3 * j + 2];
    }
    const double gs = scale[t] * d.g[t];
    const double res = scale[t] * acc + diagonal[t] * inv_radius * step[t] + gs;
    a0 += res * res; a1 += gs * gs;
  }
  s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w];}
    __syncthreads();
  }
  if (threadIdx.x == 0) {out2[0] = s0[0]; out2[1] = s1[0];}
}
void spa_launch_lin_check(const SpaDev & d, const double * scale, const double * diagonal, double inv_radius, const double * step, double * out2,
                          void * stream)
{
  hipLaunchKernelGGL(k_lin_check, dim3(1), dim3(1024), 0, (hipStream_t)stream, d, scale, diagonal, inv_radius, step, out2);
}

void spa_launch_plus(const SpaDev & d, const double * x, const double * delta, double * cand, double * out2, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  double * tmp = d.edge_lin;
  hipLaunchKernelGGL(k_plus, dim3((d.n_nodes + 255) / 256), dim3(256), 0, s, d, x, delta, cand, tmp);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, tmp, d.n_free, 1.0, out2);
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, tmp + d.n_free, d.n_free, 1.0, out2 + 1);
}

__global__ __launch_bounds__(1024) void k_grad_norms(SpaDev d, const double * x, double * out2)
{
  __shared__ double smax[1024];
  __shared__ double ssum[1024];
  double mx = 0.0, sm = 0.0;
  for (int f = threadIdx.x; f < d.n_free; f += 1024) {
    const int n = d.node_of_free[f];
    const double px = x[3 * n], py = x[3 * n + 1], pt = x[3 * n + 2];
    // x - Plus(x, -g)  (projected gradient step, trust_region_minimizer.cc)
    const double ex = px - (px - d.g[3 * f]), ey = py - (py - d.g[3 * f + 1]);
    const double et = pt - d_normalize_angle(pt - d.g[3 * f + 2]);
    mx = fmax(mx, fmax(fabs(ex), fmax(fabs(ey), fabs(et))));
    sm += px * px + py * py + pt * pt;
  }
  smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + w]);
      ssum[threadIdx.x] += ssum[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {out2[0] = smax[0]; out2[1] = ssum[0];}
}
void spa_launch_grad_norms(const SpaDev & d, const double * x, double * out2, void * stream)
{
  hipLaunchKernelGGL(k_grad_norms, dim3(1), dim3(1024), 0, (hipStream_t)stream, d, x, out2);
}

// ---------------------------------------------------------------------------------------------
// The step evaluation of an LM iteration in five launches (single GPU): fourteen small launches -- six of them one-workgroup
// sums over 30 000 values, 8-10 us each -- were 0.15 ms of every iteration.  Every kernel leaves per-workgroup partial
// sums (tree reduction in LDS, fixed order: bit-reproducible), and one last workgroup adds the partials up.
//   k_step_fused   step = -y, delta = step * scale, cand = Plus(x, delta), model-cost terms, step norms  (thread = free node)
//   k_edge_lin<true, true>  candidate cost + linearisation, cost partials
//   k_gather_H, k_gather_g<true>  normal equations at the candidate, gradient-norm partials
//   k_reduce_partials
// four lanes per free node: its row of H holds ~7 blocks, each a dependent (column -> step of the column) pair of loads; with one
// thread per node the kernel was 40 workgroups walking them one after the other (20 us for 10 000 nodes)
constexpr int kStepLanes = 4;
__global__ __launch_bounds__(256) void k_step_fused(SpaDev d, const double * __restrict__ scale, const double * __restrict__ rhs,
                                                    const double * __restrict__ x, double * step, double * delta, double * cand, double * partial)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = t / kStepLanes, sub = t % kStepLanes;
  double p[5] = {0.0, 0.0, 0.0, 0.0, 0.0};        // step.gs, step^T Hs step, non-finite marker, |x - cand|^2, |cand|^2
  const bool live = i < d.n_free;
  double acc[3] = {0.0, 0.0, 0.0};
  if (live) {
    for (int k = d.bsr_row_ptr[i] + sub; k < d.bsr_row_ptr[i + 1]; k += kStepLanes) {
      const int j = d.bsr_col[k];
      const int ej = d.elim_of_free[j];
      const double * blk = d.H + (size_t)k * 9;
      double sj[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {sj[c] = scale[3 * j + c] * -rhs[3 * ej + c];}
#pragma unroll
      for (int r = 0; r < 3; ++r) {acc[r] += blk[3 * r] * sj[0] + blk[3 * r + 1] * sj[1] + blk[3 * r + 2] * sj[2];}
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    acc[r] += __shfl_xor(acc[r], 1);
    acc[r] += __shfl_xor(acc[r], 2);
  }
  if (live && sub == 0) {
    const int e = d.elim_of_free[i];
    double st[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {st[c] = -rhs[3 * e + c];}                 // levenberg_marquardt_strategy.cc: step *= -1
    double dl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double sc = scale[3 * i + r];
      step[3 * i + r] = st[r];
      dl[r] = st[r] * sc;                                                  // trust_region_minimizer.cc: delta = step .* jacobian_scaling
      delta[3 * i + r] = dl[r];
      p[0] += st[r] * sc * d.g[3 * i + r];
      p[1] += st[r] * (acc[r] * sc);
      p[2] += (st[r] - st[r] == 0.0) ? 0.0 : 1.0;
    }
    const int n = d.node_of_free[i];
    const double px = x[3 * n], py = x[3 * n + 1], pt = x[3 * n + 2];
    const double nx = px + dl[0], ny = py + dl[1];
    const double nt = d_normalize_angle(pt + dl[2]);                       // AngleLocalParameterization, ceres_utils.h:38-55
    const double ex = px - nx, ey = py - ny, et = pt - nt;
    p[3] = ex * ex + ey * ey + et * et;
    p[4] = nx * nx + ny * ny + nt * nt;
    cand[3 * n] = nx; cand[3 * n + 1] = ny; cand[3 * n + 2] = nt;
  }
  block_reduce_store<5>(p, 0, partial + 5 * blockIdx.x);
}

// The normal equations at the candidate in ONE launch: the first nbg workgroups gather the gradient (and leave the projected-
// gradient norm partials), the others the blocks of H -- two launches in round 5, the shorter one (10 us) a boundary and a ramp
// of its own on the critical stream.
__global__ __launch_bounds__(256) void k_gather_Hg_norms(SpaDev d, const double * __restrict__ x, double * partial, int nbg)
{
  if ((int)blockIdx.x >= nbg) {gather_H(d, ((int)blockIdx.x - nbg) * 256 + (int)threadIdx.x); return;}
  const int t = blockIdx.x * 256 + threadIdx.x;
  double p[2] = {0.0, 0.0};                       // max |x - Plus(x, -g)|, |x_free|^2
  if (t < d.n_free * 3) {
    const int i = t / 3, comp = t - i * 3;
    double acc = 0.0;
    for (int k = d.node_contrib_ptr[i]; k < d.node_contrib_ptr[i + 1]; ++k) {
      const int code = d.node_contrib[k];
      const int e = code >> 1, role = code & 1;
      const double * lin = d.edge_lin + 21 * (size_t)e;
      const double * J = lin + (role ? 12 : 3);
      acc += J[comp] * lin[0] + J[3 + comp] * lin[1] + J[6 + comp] * lin[2];
    }
    d.g[t] = acc;
    const double v = x[3 * d.node_of_free[i] + comp];
    const double moved = comp == 2 ? d_normalize_angle(v - acc) : v - acc;
    p[0] = fabs(v - moved);
    p[1] = v * v;
  }
  block_reduce_store<2>(p, 1, partial + 2 * blockIdx.x);
}

// out[3..7] <- step partials, out[8] <- 0.5 * cost partials, out[9], out[10] <- gradient norm partials.  With h_out (host-coherent
// memory): the same eight numbers and the factorisation's fail word (h_out[11]) go straight to the host, then the flag h_flag <- seq
// (system-scope release): the LM loop reads its iteration's scalars without two copy nodes and a stream drain (30 us of every
// iteration).
__global__ __launch_bounds__(256) void k_reduce_partials(const double * ps, int ns, const double * pe, int ne, const double * pg, int ng, double * out,
                                                         double * h_out, int32_t * h_flag, const int32_t * fail_flag, int32_t seq)
{
  double v[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < ns; b += 256) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {v[q] += ps[5 * b + q];}
  }
  for (int b = threadIdx.x; b < ne; b += 256) {v[5] += pe[b];}
  for (int b = threadIdx.x; b < ng; b += 256) {v[6] = fmax(v[6], pg[2 * b]); v[7] += pg[2 * b + 1];}
  __shared__ double res[8];
  block_reduce_store<8>(v, 1 << 6, res);
  __syncthreads();
  if (threadIdx.x < 5) {out[3 + threadIdx.x] = res[threadIdx.x];}
  if (threadIdx.x == 5) {out[8] = 0.5 * res[5];}
  if (threadIdx.x == 6) {out[9] = res[6];}
  if (threadIdx.x == 7) {out[10] = res[7];}
  if (h_out && threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) {h_out[3 + q] = res[q];}
    h_out[8] = 0.5 * res[5]; h_out[9] = res[6]; h_out[10] = res[7];
    h_out[11] = (double)*fail_flag;
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(h_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int64_t spa_step_partials_size(const SpaDev & d)
{
  return 5 * (int64_t)((kStepLanes * d.n_free + 255) / 256) + (d.n_edges + 255) / 256 + 2 * (int64_t)((3 * d.n_free + 255) / 256) + 16;
}

// projected-gradient norm partials on their own (sharded runs: g is only complete after the all-reduce); the same
// arithmetic and the same partial layout as k_gather_g_norms
__global__ __launch_bounds__(256) void k_grad_norm_partials(SpaDev d, const double * __restrict__ x, double * partial)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  double p[2] = {0.0, 0.0};
  if (t < d.n_free * 3) {
    const int i = t / 3, comp = t - i * 3;
    const double acc = d.g[t];
    const double v = x[3 * d.node_of_free[i] + comp];
    const double moved = comp == 2 ? d_normalize_angle(v - acc) : v - acc;
    p[0] = fabs(v - moved);
    p[1] = v * v;
  }
  block_reduce_store<2>(p, 1, partial + 2 * blockIdx.x);
}

// first half: everything up to H and g of this rank's edge block [e_lo, e_hi); a sharded caller sums H || g over the ranks
// before the second half
void spa_launch_step_and_linearize(const SpaDev & cur, const SpaDev & alt, const double * scale, const double * rhs, const double * x, double * step,
                                   double * delta, double * cand, double * partial, int e_lo, int e_hi, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  const int nbs = (kStepLanes * cur.n_free + 255) / 256, nbe = (cur.n_edges + 255) / 256, nbg = (3 * cur.n_free + 255) / 256;
  double * ps = partial, * pe = ps + 5 * (size_t)nbs, * pg = pe + nbe;
  const bool sharded = e_lo > 0 || e_hi < alt.n_edges;
  hipLaunchKernelGGL(k_step_fused, dim3(nbs), dim3(256), 0, s, cur, scale, rhs, x, step, delta, cand, ps);
  if (alt.n_edges > 0) {
    // sharded: the cost is still evaluated over all edges on every rank (E threads, microseconds)
    if (sharded) {
      hipLaunchKernelGGL(k_edge_lin<false>, dim3(nbe), dim3(256), 0, s, alt, cand, 0, alt.n_edges, pe);
      hipLaunchKernelGGL(k_edge_lin<true>, dim3(nbe), dim3(256), 0, s, alt, cand, e_lo, e_hi, (double *)nullptr);
    } else {
      hipLaunchKernelGGL(k_edge_lin<true>, dim3(nbe), dim3(256), 0, s, alt, cand, e_lo, e_hi, pe);
    }
  }
  const int nbh = alt.n_edges > 0 ? (alt.n_slots * 9 + 255) / 256 : 0;
  if (sharded) {
    if (nbh > 0) {hipLaunchKernelGGL(k_gather_H, dim3(nbh), dim3(256), 0, s, alt);}
    hipLaunchKernelGGL(k_gather_g, dim3(nbg), dim3(256), 0, s, alt);
  } else {
    hipLaunchKernelGGL(k_gather_Hg_norms, dim3(nbg + nbh), dim3(256), 0, s, alt, cand, pg, nbg);
  }
}

// second half: scal[3..10] from the partial sums (sharded: the gradient norms from the summed g first)
void spa_launch_step_scalars(const SpaDev & alt, const double * cand, double * partial, bool sharded, double * scal, void * stream, double * h_out,
                             int32_t * h_flag, const int32_t * fail_flag, int32_t seq)
{
  hipStream_t s = (hipStream_t)stream;
  const int nbs = (kStepLanes * alt.n_free + 255) / 256, nbe = (alt.n_edges + 255) / 256, nbg = (3 * alt.n_free + 255) / 256;
  double * ps = partial, * pe = ps + 5 * (size_t)nbs, * pg = pe + nbe;
  if (sharded) {hipLaunchKernelGGL(k_grad_norm_partials, dim3(nbg), dim3(256), 0, s, alt, cand, pg);}
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s, ps, nbs, pe, alt.n_edges > 0 ? nbe : 0, pg, nbg, scal, h_out, h_flag, fail_flag, seq);
}

// ---------------------------------------------------------------------------------------------
// K6b: multifrontal partial Cholesky, one workgroup per front.
//
// Blocked right-looking factorisation of the front's first ns columns, panel width NB = 16:
//   (a) 16x16 diagonal block: one wave, lane = row, registers + shuffles
//   (b) panel solve X = A21 L11^-T: one thread per row; X goes back to the front (it is L21) and into
//       an LDS panel (row stride 17 doubles: conflict-free for the MFMA operand reads)
//   (c) trailing update of the lower triangle, C -= X X^T, in 16x16 tiles on the FP64 matrix cores:
//       4 x v_mfma_f64_16x16x4_f64 per tile with both operands read from the LDS panel.  The MFMA "col"
//       index (lane & 15) is mapped to the front's ROW so that every accumulator load/store of a
//       column-major front is four contiguous 128-byte segments.
// Dynamic LDS: panel of roundup16(m) rows (sized by the host for the largest front of the level).
constexpr int NB = 16;
constexpr int XS = 2 * NB + 1;               // LDS panel row stride: two 16-column panels + 1 (conflict-free MFMA operand reads)
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 16x16 diagonal block of a panel: lane l owns row l in registers, pivot columns are broadcast with
// v_readlane.  The pivot chain is the serial part of the whole factorisation, so it is kept short:
// 1/sqrt(d) from v_rsq_f64 + two Newton steps (no f64 sqrt / divide expansions), no lane masks (the
// upper-triangle garbage a lane computes is never read by another lane).  Writes L back to the front
// and L plus the reciprocal diagonal (column NB) to the LDS copy `ld`.  Rows/columns >= nb are padded
// with the identity.  Returns true when a pivot is not positive.
__device__ __noinline__ bool factor_diag_block(double * F, int m, int jb, int nb, int lane, double * ld, double * rhs_seg)
{
  double row[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    row[c] = (lane < nb && c < nb && c <= lane) ? F[(jb + lane) + (int64_t)(jb + c) * m] : ((c == lane) ? 1.0 : 0.0);
  }
  bool bad = false;
  double rdiag = 1.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double djj = readlane_f64(row[j], j);
    if (!(djj > 0.0)) {bad = true; djj = 1.0;}
    double r = __builtin_amdgcn_rsq(djj);
    const double h = 0.5 * djj;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    row[j] = (lane == j) ? djj * r : row[j] * r;
    if (lane == j) {rdiag = r;}
#pragma unroll
    for (int c = j + 1; c < NB; ++c) {
      const double lcj = readlane_f64(row[j], c);       // L[c][j]
      row[c] -= row[j] * lcj;
    }
  }
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      ld[lane * (NB + 1) + c] = (c <= lane) ? row[c] : 0.0;
      if (lane < nb && c <= lane) {F[(jb + lane) + (int64_t)(jb + c) * m] = row[c];}
    }
    ld[lane * (NB + 1) + NB] = rdiag;
  }
  // fused forward solve: y = L11^-1 b for this panel's segment of the front's right-hand side (LDS)
  {
    double v = lane < nb ? rhs_seg[lane] : 0.0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const double yj = readlane_f64(v, j) * readlane_f64(rdiag, j);
      if (lane == j) {v = yj;} else if (lane > j) {v -= row[j] * yj;}
    }
    if (lane < nb) {rhs_seg[lane] = v;}
  }
  return bad && lane < nb;
}

// Panel solve of one row: x = a L11^-T for the nb (<= 16) columns of the panel; a = gcol[c * m].  x goes
// back to the front and into the row's LDS panel slot (zero padded to 16).  nb = 0: padding row, only
// the zero fill.  `ld` = 16 x 17 LDS copy of L11 with the reciprocal diagonal in column 16.
__device__ __noinline__ void panel_row_solve(double * gcol, int m, int nb, const double * ld, double * xrow,
                                             const double * y, double * rhs_row)
{
  double xr[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {xr[c] = (c < nb) ? gcol[(int64_t)c * m] : 0.0;}
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    // The address of row c of L11 is made to depend (through an opaque zero) on x[c-2]: without it the
    // scheduler issues all 152 LDS reads of the unrolled solve up front (304 VGPRs -> spills); with it
    // at most two rows are in flight, one ahead of the data dependency x[c] <- x[c-1] that serialises
    // the columns anyway.
    int zero = 0;
    if (c >= 2) {asm volatile("v_mov_b32 %0, 0" : "=v"(zero) : "v"(__double2loint(xr[c - 2])));}
    const double * lrow = ld + c * (NB + 1) + zero;
    double v = xr[c];
#pragma unroll
    for (int q = 0; q < c; ++q) {v -= xr[q] * lrow[q];}
    xr[c] = v * lrow[NB];
  }
  double dot = 0.0;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    if (c < nb) {gcol[(int64_t)c * m] = xr[c]; dot += xr[c] * y[c];}
    xrow[c] = xr[c];
  }
  if (nb > 0) {*rhs_row -= dot;}          // fused forward solve: b_i -= L21[i, :] . y
}

constexpr int kMaxLdsRows = 568;           // 568 * 33 * 8 B = 150 KB of the CU's 160 KB LDS

// C -= X X^T on 16x16 tiles of the trailing matrix whose row/column 0 is global row `rb`.  X has KD = 16 or 32
// columns: the LDS panel (row stride XS; columns 16..31 = the second panel of a pair) or, for fronts too large
// for it, the panel columns jb .. jb+KD-1 of the front itself.  thin: only tile column 0 (what the second panel
// of a pair needs before it can be factored); otherwise tile columns >= 1 (thin must have run first), or all
// columns when `all` is set (single panel).
template <bool kLds, int KD>
__device__ __forceinline__ void trailing_update(double * F, const double * Xs, int m, int rb, int jb, int nb_a, int nb_b,
                                                int nrows, int nrows_pad, bool thin, bool all,
                                                int lane, int wave, int nwaves)
{
  constexpr int TU = 4;       // tiles in flight per wave: their accumulator loads are issued together
  const int nt = nrows_pad >> 4;
  const int ntri = all ? nt : nt - 1;             // side of the triangle of tiles walked when not thin
  const int ntiles = thin ? nt : ntri * (ntri + 1) / 2;
  const int shift = (thin || all) ? 0 : 1;
  const int lr = lane & 15, lk = lane >> 4;
  for (int t0 = wave; t0 < ntiles; t0 += nwaves * TU) {
    v4d acc[TU];
    double * cp[TU];
    int tI[TU], tJ[TU];
    bool ok[TU][4];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int t = t0 + u * nwaves;
      int I, J;
      if (thin) {
        I = t; J = 0;
      } else {
        I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (I * (I + 1) / 2 > t) {--I;}
        while ((I + 1) * (I + 2) / 2 <= t) {++I;}
        J = t - I * (I + 1) / 2;
        I += shift; J += shift;
      }
      tI[u] = I; tJ[u] = J;
      const int frow = 16 * I + lr;               // row of the trailing matrix held by this lane
      const int fcol0 = 16 * J + lk;              // its column for accumulator register 0 (+4 per register)
      cp[u] = F + (rb + frow) + (int64_t)(rb + fcol0) * m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fc = fcol0 + 4 * r;
        ok[u][r] = t < ntiles && frow < nrows && fc < nrows && fc <= frow;
        acc[u][r] = ok[u][r] ? cp[u][(int64_t)(4 * r) * m] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      if (t0 + u * nwaves >= ntiles) {continue;}       // wave-uniform
      if (kLds) {
        const double * xa = Xs + (16 * tJ[u] + lr) * XS + lk;
        const double * xb = Xs + (16 * tI[u] + lr) * XS + lk;
#pragma unroll
        for (int kk = 0; kk < KD / 4; ++kk) {
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * kk], xb[4 * kk], acc[u], 0, 0, 0);
        }
      } else {
        const int ra = 16 * tJ[u] + lr, rbw = 16 * tI[u] + lr;
#pragma unroll
        for (int kk = 0; kk < KD / 4; ++kk) {
          const int kc = 4 * kk + lk;
          // panel A = columns jb .. jb+15, panel B = jb+16 .. ; B has no rows 0..15 (its own diagonal block)
          const bool kv = kc < 16 ? kc < nb_a : (kc - 16) < nb_b;
          const double a = (ra < nrows && kv && (kc < 16 || ra >= 16)) ? F[(rb + ra) + (int64_t)(jb + kc) * m] : 0.0;
          const double b = (rbw < nrows && kv && (kc < 16 || rbw >= 16)) ? F[(rb + rbw) + (int64_t)(jb + kc) * m] : 0.0;
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, acc[u], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ok[u][r]) {cp[u][(int64_t)(4 * r) * m] = acc[u][r];}
      }
    }
  }
}

// Extend-add of a whole level, chip-wide: workgroup (front, block of 16 destination columns).  Inside k_factor the sums
// of a 456-row front's two children took one CU 65 us per level -- a quarter of the level on the narrow upper levels,
// where most of the chip idles.  A child's struct rows map to INCREASING positions in the parent (both are in
// elimination order), so the child columns that land in this workgroup's destination block are one contiguous range;
// every entry of the block is summed by this workgroup alone, children in turn: no atomics, the same order every run.
// part: 0 everything; 1 only the front's PIVOT BLOCK (destination rows and columns < ns: what k_potrf reads); 2 everything
// else (it runs on a second stream beside k_potrf of the same level).
__global__ __launch_bounds__(1024) void k_extend_add(SpaDev d, const int32_t * __restrict__ level_fronts, int part)
{
  const int nw = (int)blockDim.x >> 6;
  const int k = level_fronts[blockIdx.x];
  const int m = d.front_m[k];
  const int nsp_front = d.front_ns[k];
  if (part == 1 && 16 * (int)blockIdx.y >= nsp_front) {return;}
  const int c_lo = 16 * (int)blockIdx.y, c_hi = min(m, c_lo + 16);
  if (c_lo >= m) {return;}
  double * F = d.fronts + d.front_off[k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int32_t s_rp[2048];            // the child's struct rows' block positions in this front (m <= 8000 -> < 2667 blocks)
  // the children's descriptors first, all at once: fetched one child after the other they are a chain of six dependent
  // memory latencies per child, which was most of this kernel's 30 us
  constexpr int kBatch = 8;
  const int ci0 = d.child_ptr[k], ci1 = d.child_ptr[k + 1];
  for (int cb = ci0; cb < ci1; cb += kBatch) {
  int cid[kBatch], cm[kBatch], cns[kBatch], crp[kBatch];
  long long coff[kBatch];
#pragma unroll
  for (int q = 0; q < kBatch; ++q) {cid[q] = cb + q < ci1 ? d.child_list[cb + q] : -1;}
#pragma unroll
  for (int q = 0; q < kBatch; ++q) {
    const int c = cid[q] < 0 ? 0 : cid[q];
    cm[q] = d.front_m[c]; cns[q] = d.front_ns[c]; coff[q] = d.front_off[c]; crp[q] = d.relpos_ptr[c];
  }
#pragma unroll
  for (int q = 0; q < kBatch; ++q) {
    if (cid[q] < 0) {continue;}
    const int mc = cm[q], nsc = cns[q], nuc = mc - nsc;
    const double * Uc = d.fronts + coff[q] + nsc + (int64_t)nsc * mc;
    const int32_t * rp = d.relpos + crp[q];
    const int nblk = nuc / 3;
    const bool staged = nblk <= 2048;       // staged in LDS: the two bisections below are chains of dependent reads
    if (staged) {for (int i = threadIdx.x; i < nblk; i += blockDim.x) {s_rp[i] = rp[i];}}
    __syncthreads();
    auto pos = [&](int a) {return 3 * (staged ? s_rp[a / 3] : rp[a / 3]) + a % 3;};
    // first child column with pos >= c_lo, first with pos >= c_hi
    int lo = 0, hi = nuc;
    while (lo < hi) {const int mid = (lo + hi) >> 1; if (pos(mid) < c_lo) {lo = mid + 1;} else {hi = mid;}}
    const int b_lo = lo;
    hi = nuc;
    while (lo < hi) {const int mid = (lo + hi) >> 1; if (pos(mid) < c_hi) {lo = mid + 1;} else {hi = mid;}}
    const int b_hi = lo;
    // first child row that lands below the front's pivot block
    int a_piv = nuc;
    if (part != 0) {
      int l2 = 0, h2 = nuc;
      while (l2 < h2) {const int mid = (l2 + h2) >> 1; if (pos(mid) < nsp_front) {l2 = mid + 1;} else {h2 = mid;}}
      a_piv = l2;
    }
    for (int b = b_lo + wave; b < b_hi; b += nw) {
      double * dst = F + (int64_t)pos(b) * m;
      const double * src = Uc + (int64_t)b * mc;
      const bool pivot_col = b < a_piv;
      const int a_begin = part == 2 && pivot_col ? max(b, a_piv) : b;
      const int a_end = part == 1 ? (pivot_col ? a_piv : 0) : nuc;
      for (int a0 = a_begin; a0 < a_end; a0 += 256) {
        // four rows per lane in flight: unconditional loads at a clamped row (as `a < a_end ? load : 0` every load sat in a branch
        // of its own with a wait behind it: twelve dependent round trips per 256 rows where two do)
        double u[4], f[4];
        int pa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ac = min(a0 + lane + 64 * q, a_end - 1);
          pa[q] = pos(ac);
          u[q] = src[ac];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {f[q] = dst[pa[q]];}
#pragma unroll
        for (int q = 0; q < 4; ++q) {if (a0 + lane + 64 * q < a_end) {dst[pa[q]] = f[q] + u[q];}}
      }
    }
    __syncthreads();       // the next child may add into the same entries (and restages s_rp)
  }
  }
}

void spa_launch_extend_add(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, void * stream, int32_t part, int32_t max_ns)
{
  if (n <= 0) {return;}
  if (part == 1) {max_m = std::min(max_m, max_ns);}      // only the column blocks of the pivot block have work
  // one wave per destination column of the block of 16 (with four waves a workgroup walked its columns four at a time, each
  // walk a round trip to memory); wide levels keep the small workgroups
  static const int ea_threads = std::getenv("KH_SPA_EA_THREADS") ? std::atoi(std::getenv("KH_SPA_EA_THREADS")) : 0;
  const int threads = ea_threads ? ea_threads : ((int64_t)n * ((max_m + 15) / 16) > 512 ? 256 : 1024);
  hipLaunchKernelGGL(k_extend_add, dim3(n, (max_m + 15) / 16), dim3(threads), 0, (hipStream_t)stream, d, level_fronts, (int)part);
}

// ---- workgroup-level hand-offs between the workgroups that share one front (agent scope: the L1 of a CU is never
// refreshed by another CU's stores, MI355X_MICROARCH.md "inter-workgroup visibility") ----
// publish: everything this workgroup has stored becomes visible, then the word moves
__device__ __forceinline__ void wg_publish_add(int * word, int value)
{
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// wait until *word >= value, then make the publishers' stores visible to every thread of this workgroup.  The spin is
// bounded: a hand-off that never comes (it cannot, all workgroups of a launch are resident) must not hang the device.
__device__ __forceinline__ bool wg_wait_ge(int * word, int value)
{
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 0;
    for (int spin = 0; spin < (1 << 26); ++spin) {
      if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= value) {ok = 1; break;}
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

// G workgroups per front (G = 1 on the wide levels).  Workgroup 0 of a front is its LEADER and runs the factorisation
// as before; on the narrow upper levels, where a handful of large fronts would leave most of the chip idle, G - 1 helper
// workgroups take destination-column slices of the extend-add and tile slices of every K = 32 trailing update (the bulk
// of the flops), reading the solved panel from the front itself.  Hand-offs: sync[4 * slot + 0] extend-add arrivals,
// + 1 panel pairs published by the leader, + 2 helper completions (zeroed by the host before every factorisation).
__global__ __launch_bounds__(1024) void k_factor(SpaDev d, const int32_t * __restrict__ level_fronts, int32_t * fail_flag, long long * tbuf,
              double * rhs, double * upd, int lds_rows, int G, int * sync, int matrix_added)
{
  // KH_SPA_TIMING=1: stage timestamps (100 MHz wall clock) of the level's largest front, printed by the host
  int tcount = 0;
#define TSTAMP() do { if (tbuf && blockIdx.x == 0 && threadIdx.x == 0 && tcount < 60) {tbuf[1 + tcount++] = wall_clock64();} } while (0)
  TSTAMP();
  const int slot = (int)blockIdx.x / G, g = (int)blockIdx.x - slot * G;
  int * sy = sync + 4 * slot;
  const int k = level_fronts[slot];
  const int m = d.front_m[k], ns = d.front_ns[k];
  double * F = d.fronts + d.front_off[k];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  extern __shared__ double smem[];
  double * Xs = smem;                       // [rows below the first panel of a pair, padded to 16][XS]
  const bool use_lds = m <= kMaxLdsRows;    // larger fronts read the panel back from the front itself (L2)
  __shared__ double Ld[NB][NB + 1];
  __shared__ int s_fail;
  // Fused forward solve (K6c, first half): the front's slice of the right-hand side lives in LDS next to the
  // panel -- pivots from rhs (elimination order), struct rows accumulate the children's contributions -- and
  // every panel applies y = L11^-1 b, b_below -= L21 y while L11 and L21 are still on chip.
  double * sb = use_lds ? smem + (size_t)lds_rows * XS : smem;
  const int first = 3 * d.front_first[k];
  for (int t = tid; t < m; t += nthreads) {sb[t] = t < ns ? rhs[first + t] : 0.0;}
  __syncthreads();

  // 1. extend-add the children's update matrices (lower triangles).  Children are taken one after the
  //    other (two children may add into the same entry); within a child every wave takes columns
  //    four at a time with all loads issued before the first add, so the pass is bound by bandwidth and
  //    not by a chain of dependent L2 round trips.  The scalar row positions inside this front
  //    (3 * relpos + component) are staged in LDS once per child.
  int * pos = use_lds ? reinterpret_cast<int *>(smem) : reinterpret_cast<int *>(smem + m);
  for (int ci = d.child_ptr[k]; ci < d.child_ptr[k + 1]; ++ci) {
    const int c = d.child_list[ci];
    const int mc = d.front_m[c], nsc = d.front_ns[c], nuc = mc - nsc;
    const double * Uc = d.fronts + d.front_off[c] + nsc + (int64_t)nsc * mc;
    const int32_t * rp = d.relpos + d.relpos_ptr[c];
    for (int a = tid; a < nuc; a += nthreads) {pos[a] = 3 * rp[a / 3] + a % 3;}
    __syncthreads();
    if (g == 0) {
      const double * uc = upd + 3 * (int64_t)d.front_rows_ptr[c];      // the child's forward-solve contribution
      for (int a = tid; a < nuc; a += nthreads) {sb[pos[a]] += uc[a];}
    }
    constexpr int CB = 8;
    // matrix_added: k_extend_add has already summed the children's update matrices into the front, chip-wide
    for (int b0 = wave * CB; b0 < nuc && !matrix_added; b0 += nwaves * CB) {
      for (int a0 = b0; a0 < nuc; a0 += 64) {
        const int a = a0 + lane;
        double u[CB], f[CB];
        double * dst[CB];
        bool on[CB];
        const int pa = a < nuc ? pos[a] : 0;
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          const int b = b0 + q;
          // with several workgroups on the front, a DESTINATION column (block of 16) has one owner: whatever the child,
          // all additions into an entry come from the same workgroup, children in turn
          on[q] = a < nuc && b < nuc && a >= b && (G == 1 || ((pos[b] >> 4) % G) == g);
          dst[q] = F + pa + (int64_t)(on[q] ? pos[b] : 0) * m;
          u[q] = on[q] ? Uc[a + (int64_t)b * mc] : 0.0;
          f[q] = on[q] ? *dst[q] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < CB; ++q) {if (on[q]) {*dst[q] = f[q] + u[q];}}
      }
    }
    __syncthreads();
  }

  TSTAMP();
  if (G > 1) {
    wg_publish_add(&sy[0], 1);
    if (g > 0) {
      // helper: the same walk over the panels as the leader's below, taking part in the K = 32 updates only
      int pairs = 0;
      for (int jb = 0; jb < ns; ) {
        const int nb_a = min(NB, ns - jb);
        const int rb = jb + nb_a;
        const int nrows = m - rb;
        const int nrows_pad = (nrows + 15) & ~15;
        if (jb + 2 * NB > ns) {jb += NB; continue;}
        ++pairs;
        if (!wg_wait_ge(&sy[1], pairs)) {if (tid == 0) {atomicExch(fail_flag, 1);} return;}
        if (use_lds) {
          // the solved panel pair into this workgroup's LDS (rows relative to rb; the second panel has no rows 0..15),
          // coalesced along the rows, then the same LDS-operand MFMA path as the leader's
          for (int t = tid; t < nrows_pad * 2 * NB; t += nthreads) {
            const int c = t / nrows_pad, row = t - c * nrows_pad;
            Xs[row * XS + c] = (row < nrows && (c < NB || row >= NB)) ? F[(rb + row) + (int64_t)(jb + c) * m] : 0.0;
          }
          __syncthreads();
          trailing_update<true, 32>(F, Xs, m, rb, jb, nb_a, NB, nrows, nrows_pad, false, false, lane, g * nwaves + wave, G * nwaves);
        } else {
          trailing_update<false, 32>(F, nullptr, m, rb, jb, nb_a, NB, nrows, nrows_pad, false, false, lane, g * nwaves + wave, G * nwaves);
        }
        wg_publish_add(&sy[2], 1);
        jb += 2 * NB;
      }
      return;
    }
    if (!wg_wait_ge(&sy[0], G)) {if (tid == 0) {atomicExch(fail_flag, 1);} return;}
  }
  // 2. blocked right-looking partial Cholesky of the first ns columns
  int pairs_published = 0;
  if (tid == 0) {s_fail = 0;}
  __syncthreads();
  // Panels are taken in pairs with a delayed update: after panel A only the next 16 columns of the trailing
  // matrix are updated (thin), panel B is factored, and then BOTH panels are applied to the rest in one pass
  // with K = 32 -- half as many sweeps over the trailing matrix, twice the flops per accumulator load/store.
  auto panel = [&](int jb, int nb, int rb, double * xcols) {
    // (a) diagonal block: wave 0
    if (wave == 0) {
      if (factor_diag_block(F, m, jb, nb, lane, &Ld[0][0], sb + jb)) {s_fail = 1;}
    }
    __syncthreads();
    TSTAMP();
    // (b) panel: X = F[rows, jb:jb+nb] * Ld^{-T}, rows below the diagonal block; one thread per row (out of
    //     line, see panel_row_solve); fronts too large for the LDS panel solve in place in the front instead
    const int r0 = jb + nb;
    const int nrows = m - r0;
    const int nrows_pad = (nrows + 15) & ~15;
#pragma unroll 1
    for (int i = tid; i < nrows_pad; i += nthreads) {
      double * gcol = F + (r0 + i) + (int64_t)jb * m;       // F[r0 + i][jb + c] = gcol[c * m]
      if (use_lds) {
        panel_row_solve(gcol, m, i < nrows ? nb : 0, &Ld[0][0], xcols + (size_t)(r0 - rb + i) * XS, sb + jb,
                        sb + min(r0 + i, m - 1));
      } else if (i < nrows) {
        double dot = 0.0;
#pragma unroll 1
        for (int c = 0; c < nb; ++c) {
          const double * lrow = &Ld[c][0];
          double v = gcol[(int64_t)c * m];
#pragma unroll 4
          for (int q = 0; q < c; ++q) {v -= gcol[(int64_t)q * m] * lrow[q];}
          v *= lrow[NB];
          gcol[(int64_t)c * m] = v;
          dot += v * sb[jb + c];
        }
        sb[r0 + i] -= dot;
      }
    }
    __syncthreads();
    TSTAMP();
  };
  for (int jb = 0; jb < ns; ) {
    const int nb_a = min(NB, ns - jb);
    const int rb = jb + nb_a;                 // first row / column of the trailing matrix of this pair
    const int nrows = m - rb;
    const int nrows_pad = (nrows + 15) & ~15;
    // only two FULL panels pair up: the thin update covers exactly the 16 columns of the second panel, so a
    // partial second panel would leave the non-pivot columns of that tile column without its contribution
    const bool pair = jb + 2 * NB <= ns;
    if (use_lds) {
      // the second panel has no rows 0..15 (they are its diagonal block): zero that corner once per pair
      for (int t = tid; t < 16 * NB; t += nthreads) {Xs[(t >> 4) * XS + NB + (t & 15)] = 0.0;}
    }
    panel(jb, nb_a, rb, Xs);
    if (!pair) {
      if (use_lds) {
        trailing_update<true, 16>(F, Xs, m, rb, jb, nb_a, 0, nrows, nrows_pad, false, true, lane, wave, nwaves);
      } else {
        trailing_update<false, 16>(F, Xs, m, rb, jb, nb_a, 0, nrows, nrows_pad, false, true, lane, wave, nwaves);
      }
      __syncthreads();
      TSTAMP();
      jb += NB;
      continue;
    }
    if (use_lds) {
      trailing_update<true, 16>(F, Xs, m, rb, jb, nb_a, 0, nrows, nrows_pad, true, false, lane, wave, nwaves);
    } else {
      trailing_update<false, 16>(F, Xs, m, rb, jb, nb_a, 0, nrows, nrows_pad, true, false, lane, wave, nwaves);
    }
    __syncthreads();
    const int nb_b = NB;
    panel(jb + NB, nb_b, rb, Xs + NB);
    if (G > 1) {wg_publish_add(&sy[1], 1); ++pairs_published;}      // both panels are in the front: the helpers may start
    if (use_lds) {
      trailing_update<true, 32>(F, Xs, m, rb, jb, nb_a, nb_b, nrows, nrows_pad, false, false, lane, wave, G * nwaves);
    } else {
      trailing_update<false, 32>(F, Xs, m, rb, jb, nb_a, nb_b, nrows, nrows_pad, false, false, lane, wave, G * nwaves);
    }
    if (G > 1) {
      if (!wg_wait_ge(&sy[2], (G - 1) * pairs_published)) {if (tid == 0) {atomicExch(fail_flag, 1);} return;}
    }
    __syncthreads();
    TSTAMP();
    jb += 2 * NB;
  }
  // forward-solve results: y of the pivots back to rhs, the struct rows' partial sums to this front's slot
  for (int t = tid; t < ns; t += nthreads) {rhs[first + t] = sb[t];}
  {
    double * uk = upd + 3 * (int64_t)d.front_rows_ptr[k];
    for (int q = tid; q < m - ns; q += nthreads) {uk[q] = sb[ns + q];}
  }
  if (tbuf && blockIdx.x == 0 && threadIdx.x == 0) {tbuf[0] = tcount; tbuf[63] = ((long long)m << 32) | ns;}
  if (tid == 0 && s_fail) {atomicExch(fail_flag, 1);}
}

// ---------------------------------------------------------------------------------------------
// lane n of every row of 16 lanes, to all lanes of that row: two v_mov_b32_dpp row_newbcast (full-rate VALU; a v_readlane
// pair goes through SGPRs and stalls on the VALU->SGPR->VALU hazards: 30 of them per pivot were most of a pivot's 230 ns)
template <int N>
__device__ __forceinline__ double row_bcast_c(double v)
{
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double row_bcast(double v, int n)      // n is a constant after unrolling
{
  switch (n) {
    case 0: return row_bcast_c<0>(v); case 1: return row_bcast_c<1>(v); case 2: return row_bcast_c<2>(v); case 3: return row_bcast_c<3>(v);
    case 4: return row_bcast_c<4>(v); case 5: return row_bcast_c<5>(v); case 6: return row_bcast_c<6>(v); case 7: return row_bcast_c<7>(v);
    case 8: return row_bcast_c<8>(v); case 9: return row_bcast_c<9>(v); case 10: return row_bcast_c<10>(v); case 11: return row_bcast_c<11>(v);
    case 12: return row_bcast_c<12>(v); case 13: return row_bcast_c<13>(v); case 14: return row_bcast_c<14>(v); default: return row_bcast_c<15>(v);
  }
}


void spa_launch_factor_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, int32_t max_ns, int32_t * fail_flag,
                             double * rhs, double * upd, double * fsb, int32_t * sync, int32_t matrix_added, void * stream)
{
  // The any-size form: what factorises a level whose fronts do not fit the level pipeline's LDS (spa_level_pipeline_fits: more
  // than ~2700 rows), and what `kh_spa_set_debug` factor mode 1 / 2 selects so that the tests can lay an independent
  // factorisation beside the pipeline's.  One workgroup per front (the round-2 panel-pair kernels, a subset of the pipeline's
  // domain, are gone).
  (void)max_ns; (void)fsb;
  if (n <= 0) {return;}
  const int G = 1;
  const int threads = max_m <= 96 ? 256 : (max_m <= 192 ? 512 : 1024);
  static long long * tbuf = nullptr;
  static const bool timing = std::getenv("KH_SPA_TIMING") != nullptr;
  if (timing && !tbuf) {(void)hipHostMalloc(reinterpret_cast<void **>(&tbuf), 64 * sizeof(long long), hipHostMallocDefault);}
  const int lds_rows = ((max_m < kMaxLdsRows ? max_m : kMaxLdsRows) + 15) & ~15;
  // panel + the front's right-hand side behind it; fronts beyond the LDS panel keep only the rhs (+ the
  // extend-add position table) in LDS
  const int small_m = max_m < kMaxLdsRows ? max_m : kMaxLdsRows;
  const size_t lds = sizeof(double) * std::max((size_t)lds_rows * XS + small_m, (size_t)max_m + (max_m + 1) / 2 + 8);
  static std::atomic<unsigned long long> attr_done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(k_factor), (int)(sizeof(double) * ((size_t)kMaxLdsRows * XS + kMaxLdsRows + 16)), attr_done);
  hipLaunchKernelGGL(k_factor, dim3(n * G), dim3(threads), lds, (hipStream_t)stream, d, level_fronts, fail_flag, timing ? tbuf : nullptr, rhs, upd,
                     lds_rows, G, sync, (int)matrix_added);
  if (timing) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    std::fprintf(stderr, "[k_factor] n=%d max_m=%d front0 m=%lld ns=%lld stamps(x10ns):", n, max_m, tbuf[63] >> 32, tbuf[63] & 0xffffffff);
    for (int i = 1; i < (int)tbuf[0]; ++i) {std::fprintf(stderr, " %lld", tbuf[1 + i] - tbuf[i]);}
    std::fprintf(stderr, "\n");
  }
}

// ---------------------------------------------------------------------------------------------
// K6c: backward triangular solve, one workgroup (256 threads) per front, blocked by 16 like the
// factorisation (the forward solve is fused into k_factor: a front gathers its children's contributions --
// upd, one segment per front = its struct rows -- instead of scattering with atomics, so the solve is
// bit-reproducible).  rhs is in elimination order.  Dynamic LDS: m doubles.
__device__ __forceinline__ void load_diag_block(const double * F, int m, int jb, int nb, int lane, double (&col)[NB])
{
  // lane c holds column c of the 16x16 diagonal block: col[j] = L[jb+j][jb+c] (j >= c)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    col[j] = (lane < nb && j < nb && j >= lane) ? F[(jb + j) + (int64_t)(jb + lane) * m] : ((j == lane) ? 1.0 : 0.0);
  }
}

// 1024 threads: the 16 columns of a block take their dot products at once (one wave each) instead of in four rounds --
// the sweep is a chain of (dot, 16 x 16 back substitution) steps, each a memory latency long, so the rounds were the time
__global__ __launch_bounds__(1024) void k_backward(SpaDev d, const int32_t * __restrict__ level_fronts, double * rhs)
{
  const int k = level_fronts[blockIdx.x];
  const int m = d.front_m[k], ns = d.front_ns[k], nu = m - ns;
  const double * F = d.fronts + d.front_off[k];
  const int first = 3 * d.front_first[k];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
  extern __shared__ double sb[];
  const int32_t * rows = d.front_rows + d.front_rows_ptr[k];
  for (int t = tid; t < ns; t += nthreads) {sb[t] = rhs[first + t];}
  for (int q = tid; q < nu; q += nthreads) {sb[ns + q] = rhs[3 * rows[q / 3] + q % 3];}
  __syncthreads();
  const int nblk = (ns + NB - 1) / NB;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int jb = blk * NB;
    const int nb = min(NB, ns - jb);
    const int r0 = jb + nb;
    // w[t] = y[t] - sum_{i >= r0} L[i][t] x[i] : one wave per column, lanes over the rows
    for (int c = wave; c < nb; c += nwaves) {
      const double * col = F + (int64_t)(jb + c) * m;
      double acc = 0.0;
      for (int i = r0 + lane; i < m; i += 64) {acc += col[i] * sb[i];}
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {acc += __shfl_xor(acc, s);}
      if (lane == 0) {sb[jb + c] -= acc;}
    }
    __syncthreads();
    if (wave == 0) {
      double col[NB];
      load_diag_block(F, m, jb, nb, lane, col);
      double v = lane < nb ? sb[jb + lane] : 0.0;
#pragma unroll
      for (int j = NB - 1; j >= 0; --j) {
        const double xj = readlane_f64(v, j) / readlane_f64(col[j], j);   // lane j: col[j] = L[jb+j][jb+j]
        if (lane == j) {v = xj;} else if (lane < j) {v -= col[j] * xj;}    // lane c < j: col[j] = L[jb+j][jb+c]
      }
      if (lane < nb) {sb[jb + lane] = v;}
    }
    __syncthreads();
  }
  for (int t = tid; t < ns; t += nthreads) {rhs[first + t] = sb[t];}
}

void spa_launch_backward_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t max_m, double * rhs, void * stream)
{
  if (n <= 0) {return;}
  hipLaunchKernelGGL(k_backward, dim3(n), dim3(1024), sizeof(double) * max_m, (hipStream_t)stream, d, level_fronts, rhs);
}


// =============================================================================================
// K6, round 3: the level pipeline  potrf -> trsm -> syrk  (one launch each per level of the assembly tree).
//
// The panel-pair kernels above keep a front on ONE compute unit from its first pivot to its last: the pivot chain, the
// row solves and the K = 32 updates of all its panel pairs run back to back, and on the upper levels of the tree -- a
// handful of fronts with 100+ pivots and 300-400 rows each -- 250 of the 256 CUs idle while those few grind through 6-7
// pairs (rocprof, round 2: 34 k_factor2 + 24 k_update2 launches = 1.8 ms per factorisation at 0.5 TFLOP/s).  The only
// inherently serial part of a front is the Cholesky of its ns x ns pivot block; everything else is matrix products:
//   potrf  one workgroup per front: F11 = L11 L11^T inside LDS (ns <= 128: the symbolic analysis splits larger
//          supernodes into chains), and in the SAME sweep W = L11^-T: the identity rides along as extra rows of the panel
//          (x L11^T = e_i, solved block column by block column like every other row), stored in the unused upper triangle
//          of the LDS matrix.  The 16 x 16 inverse of a diagonal block comes out of its pivot chain the same way (the rows
//          of the identity reuse the chain's own broadcasts), so every row solve of the panel is a 16 x 16 x 16 MFMA
//          product instead of a substitution.  y1 = L11^-1 b1 (forward solve) is one product with W^T.
//   trsm   L21 = F21 W, a plain GEMM over row slabs of every front of the level (chip-wide), fused with the forward-solve
//          contribution  b2 -= L21 y1.
//   syrk   F22 -= L21 L21^T over 32 x 32 / 64 x 64 tiles of every front of the level (chip-wide).
// With W the backward solve of a level is two matrix-vector products (k_backward3) instead of a chain of 16 x 16
// substitutions, each a memory latency long.  Fronts are numbered level by level, so a workgroup finds its front at
// first_front + blockIdx.x and everything it needs to know about it in ONE 64-byte descriptor: on this problem size a
// dependent global load is about a microsecond, and a chain of four index look-ups was a third of a small kernel's time.
constexpr int kPotrfMaxNs = 128;
constexpr int XDS = NB + 2;                // row stride of the 16 x 16 inverse of a diagonal block in LDS

// 16 x 16 Cholesky in the registers of lanes 0..15 (lane = row) with the identity riding along: xr = row `lane` of the
// identity on entry, row `lane` of L^-T on return (x L^T = e, right-looking: the L[c][j] a pivot broadcasts for the
// trailing update of the block are exactly the factors the rows of the identity need).
__device__ __forceinline__ bool diag_chain_inv(double (&row)[NB], double (&xr)[NB], int lane, double & rdiag)
{
  bool bad = false;
  rdiag = 1.0;
  const int l16 = lane & 15;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double djj = row_bcast(row[j], j);
    if (!(djj > 0.0)) {bad = true; djj = 1.0;}
    double r = __builtin_amdgcn_rsq(djj);
    const double h = 0.5 * djj;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    row[j] = (l16 == j) ? djj * r : row[j] * r;
    if (l16 == j) {rdiag = r;}
    xr[j] *= r;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) {
      const double lcj = row_bcast(row[j], c);          // L[c][j]
      row[c] -= row[j] * lcj;
      xr[c] -= xr[j] * lcj;
    }
  }
  return bad;
}

// Diagonal block `blk` of the LDS matrix (row stride LD): L over its lower part, the strictly upper part of L^-T over its
// strictly upper part, all of L^-T to xd (16 x XDS), the reciprocal pivots (= diagonal of L^-T) to rdv
__device__ __forceinline__ bool potrf_diag(double * blk, int LD, int lane, double * xd, double * rdv)
{
  double row[NB], xr[NB];
  const int l16 = lane & 15;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    row[c] = (lane < NB && c <= lane) ? blk[lane * LD + c] : ((lane >= NB && c == l16) ? 1.0 : 0.0);
    xr[c] = c == l16 ? 1.0 : 0.0;
  }
  double rdiag;
  const bool bad = diag_chain_inv(row, xr, lane, rdiag);
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      blk[lane * LD + c] = (c <= lane) ? row[c] : xr[c];
      xd[lane * XDS + c] = xr[c];
    }
    rdv[lane] = rdiag;
  }
  return bad && lane < NB;
}

// ---- the same diagonal block on the matrix cores, four pivots at a time (round 6) ----
// The chain above issues ~670 double-precision instructions for 16 pivots on a wave of which a quarter of the lanes work
// (3.2 us, measured: the floor under every level of the factorisation).  Here the 16 x 16 tile never leaves the accumulator
// layout of v_mfma_f64_16x16x4 -- lane (lr, lk) holds M[lr][lk + 4 r] in register r, i.e. register p IS column block p and
// its entry is exactly what the lane has to supply, as the A and as the B operand, for the rank-4 update of the columns
// behind block p:
//   gather   the 4 x 4 diagonal block of column block p: one ds_write of register p (64 doubles of scratch), ten uniform reads
//   chol4    its Cholesky factor and T = L_pp^-T in uniform registers (every lane the same ~60 operations)
//   panel    M[:, 4p .. 4p+3] <- M[:, 4p .. 4p+3] T   ONE MFMA: b = register p, a = T[lk][lr - 4p] on the lanes lr of block p
//   update   M[:, behind] -= panel panel^T             ONE MFMA: a = b = the new register p
// and the identity that rides along (E -> L^-T) takes the same two MFMAs with its own register p as b.  Entries above the
// diagonal hold garbage that stays in registers nobody reads (rows above block p only feed rows above block p); the rows
// of the diagonal block itself come out of `chol4`, not of the panel product.  ~700 clocks per four pivots.
// One over the square root to full double precision: v_rsq_f64 is good to ~2^-26 or better (tools/diag_probe.hip prints its
// worst error), so one THIRD-order step r (1 + e/2 + 3 e^2/8), e = 1 - d r^2, lands below 2^-70: five dependent operations
// where two Newton steps take seven.
__device__ __forceinline__ double rsqrt_full(double d)
{
  const double r = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-d * r, r, 1.0);
  const double c = __builtin_fma(e, 0.375, 0.5) * e;
  return __builtin_fma(r, c, r);
}

// T = L^-T of the 4 x 4 block a (a[i][j], j <= i), as v = L^-1 (lower): the factor itself is not formed -- nobody reads the
// diagonal tile's L (the panel product leaves the rows below it, and every consumer of the pivot block works with L^-T).
// A pivot that is not positive is reported; its reciprocal square root (NaN or infinite) flows on, the factorisation has failed
// and every number behind it is flagged by the caller's fail word.  `mid` runs after the second pivot: the place where the
// caller parks matrix-core work that must not sit between the pivots' dependent operations and their issue slots.
template <typename Mid>
__device__ __forceinline__ bool chol4_inverse(double (&m)[4][4], double (&v)[4][4], Mid mid)
{
  double r[4], l[4][4];
  double dmin = m[0][0];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j == 2) {mid();}
    dmin = fmin(dmin, m[j][j]);
    r[j] = rsqrt_full(m[j][j]);
#pragma unroll
    for (int i = j + 1; i < 4; ++i) {l[i][j] = m[i][j] * r[j];}
#pragma unroll
    for (int i = j + 1; i < 4; ++i) {
#pragma unroll
      for (int c = j + 1; c <= i; ++c) {m[i][c] -= l[i][j] * l[c][j];}
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i][i] = r[i];
#pragma unroll
    for (int j = i - 1; j >= 0; --j) {
      double acc = l[i][j] * v[j][j];
#pragma unroll
      for (int k = j + 1; k < i; ++k) {acc += l[i][k] * v[k][j];}
      v[i][j] = -acc * r[i];
    }
  }
  return dmin > 0.0;
}

// tile: the diagonal tile in accumulator layout (lower part meaningful, identity on the padding).  Outputs: the strictly upper
// part of L^-T over the strictly upper part of blk (row stride LD; the lower triangle of blk, L itself, is left alone), all of
// L^-T to xd (16 x XDS), the reciprocal pivots (= diagonal of L^-T) to rdv.  sc: 64 doubles of LDS scratch owned by this wave.
// The identity's two matrix-core operations of block p are issued inside block p + 1 -- the panel product while the next
// block's gather is on its way through LDS, the update between its second and third pivot -- so that the matrix pipe is free
// when the tile's own panel and update arrive: those two are the only ones the next pivot waits for.
#ifdef KH_DIAG_STAMPS
#define DSTAMP(i, v) do { asm volatile("s_nop 0" : "+v"(v)); long long t_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
  asm volatile("s_nop 0" : "+v"(v)); g_dstamp[i] = t_; } while (0)
__device__ long long g_dstamp[32];
#else
#define DSTAMP(i, v) do {} while (0)
#endif
__device__ __forceinline__ bool potrf_diag_mfma(v4d tile, double * blk, int LD, int lane, double * xd, double * rdv, double * sc)
{
  const int lr = lane & 15, lk = lane >> 4;
  const int c4 = lr & 3, b4 = lr >> 2;
  v4d E;
#pragma unroll
  for (int r = 0; r < 4; ++r) {E[r] = (lk + 4 * r == lr) ? 1.0 : 0.0;}
  bool bad = false;
  const v4d zero = v4d{0.0, 0.0, 0.0, 0.0};
  double ta_prev = 0.0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    // gather the diagonal 4 x 4 block: sc[lr + 16 lk] = M[lr][4 p + lk]
    DSTAMP(5 * p + 0, tile[p]);
    sc[lane] = tile[p];
    v4d en = zero;
    if (p > 0) {en = __builtin_amdgcn_mfma_f64_16x16x4f64(ta_prev, E[p - 1], zero, 0, 0, 0);}
    double a[4][4], v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j <= i; ++j) {a[i][j] = sc[(4 * p + i) + 16 * j];}
    }
    DSTAMP(5 * p + 1, a[3][3]);
    __builtin_amdgcn_sched_barrier(0);
    const bool ok = chol4_inverse(a, v, [&]() {
      __builtin_amdgcn_sched_barrier(0);
      if (p > 0) {
        E[p - 1] = en[p - 1];
        const v4d eu = __builtin_amdgcn_mfma_f64_16x16x4f64(-tile[p - 1], E[p - 1], E, 0, 0, 0);
#pragma unroll
        for (int r = p; r < 4; ++r) {E[r] = eu[r];}
      }
    });
    if (!ok) {bad = true;}
    DSTAMP(5 * p + 2, v[3][0]);
    // T[k][c] = v[c][k] on lane (lr = 4 p + c, lk = k), zero elsewhere
    double tsel = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int k = 0; k <= c; ++k) {tsel = (c4 == c && lk == k) ? v[c][k] : tsel;}
    }
    double ta = (b4 == p) ? tsel : 0.0;
    DSTAMP(5 * p + 3, ta);
    const v4d pn = __builtin_amdgcn_mfma_f64_16x16x4f64(ta, tile[p], zero, 0, 0, 0);
    tile[p] = pn[p];
    DSTAMP(5 * p + 4, tile[p]);
    if (p < 3) {
      const v4d tu = __builtin_amdgcn_mfma_f64_16x16x4f64(-tile[p], tile[p], tile, 0, 0, 0);
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {tile[r] = tu[r];}
    }
    ta_prev = ta;
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const v4d en = __builtin_amdgcn_mfma_f64_16x16x4f64(ta_prev, E[3], zero, 0, 0, 0);
    E[3] = en[3];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = lk + 4 * r;
    if (c > lr) {blk[lr * LD + c] = E[r];}
    xd[lr * XDS + c] = E[r];
    if (c == lr) {rdv[lr] = E[r];}
  }
  return bad;
}

// row "solve" of one 16-row tile: T <- T Dinv^T  (Dinv^T = L_jj^-T in xd), in place, one wave
__device__ __forceinline__ v4d potrf_rowsolve_tile(double * tile, int LD, const double * xd, int lane)
{
  const int lr = lane & 15, lk = lane >> 4;
  const double * tb = tile + lr * LD + 4 * lk;         // b: T[lr][k]
  const double * xa = xd + 4 * lk * XDS + lr;          // a: (L^-T)[k][lr]
  double b[4], a[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {b[kk] = tb[kk]; a[kk] = xa[kk * XDS];}
  v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);}
  double * cp = tile + lr * LD + lk;
#pragma unroll
  for (int r = 0; r < 4; ++r) {cp[4 * r] = acc[r];}
  return acc;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, and k_potrf's waves
// keep the NEXT step's tile loads in flight across a step -- with the full barrier every step ended by sitting out a memory
// latency (1-2 us of a 3 us step)
__device__ __forceinline__ void lds_barrier() {asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");}

// KH_SPA_TIMING: shader-clock stamps of workgroup 0's thread 0, kept in LDS (behind the kernel's own dynamic LDS) and written out by
// the kernel's last statement -- stamps stored straight to host memory put a PCIe round trip into the next vector-memory wait
#define PSTAMP() do { if (tbuf && blockIdx.x == 0 && threadIdx.x == 0 && tcount < 60) {stamps[tcount++] = clock64();} } while (0)

// ---- children's update matrices, read in place (no extend-add pass) ----
// child number s of a front: the first three sit in the front's own descriptor, further ones (rare: a separator above
// several disconnected pieces) come from the child list
__device__ __forceinline__ ChildInfo child_info(const SpaDev & d, const FrontDesc & fd, int s)
{
  if (s == 0) {return fd.ch[0];}
  if (s == 1) {return fd.ch[1];}
  if (s == 2) {return fd.ch[2];}
  const FrontDesc & c = d.desc[d.child_list[fd.child_ptr + s]];
  ChildInfo ci;
  ci.off = c.off; ci.m = c.m; ci.ns = c.ns; ci.rows_ptr = c.rows_ptr; ci.pad = 0;
  return ci;
}

// round 6 (Symbolic::scatter_mode): children read in place come first in the child list
__device__ __forceinline__ int front_nkids(const SpaDev & d, const FrontDesc & fd)
{
  return d.gather ? fd.child_end - fd.child_ptr : (d.scatter ? (fd.flags >> 8) : 0);
}
// buffer B of a front, or nullptr when no child adds into it
__device__ __forceinline__ const double * front_b(const SpaDev & d, const FrontDesc & fd)
{
  return (d.scatter && (fd.flags & 4)) ? d.fronts_b + fd.off : nullptr;
}

// entry (i, j), i >= j (scalar positions in the parent front), of one child's update matrix, 0 where the child has none
__device__ __forceinline__ double gather_entry(const double * __restrict__ fronts, const ChildInfo & c, const int32_t * __restrict__ inv, int i, int j)
{
  const int in = i / 3, jn = j / 3;
  const int a = inv[in], b = inv[jn];
  const bool ok = a >= 0 && b >= 0;
  const int64_t at = ok ? (int64_t)(c.ns + 3 * a + (i - 3 * in)) + (int64_t)(c.ns + 3 * b + (j - 3 * jn)) * c.m : 0;
  const double v = fronts[c.off + at];
  return ok ? v : 0.0;
}

// Initial value of the 16 x 16 tile (rows row0.., columns col0..) of the pivot block in MFMA accumulator layout: the
// front's own (assembled) entries plus the children's update matrices; identity on the padding; entries above the
// diagonal are not meaningful.  In two halves: pivot_tile_load only REQUESTS the front's entries (nothing in it touches a
// loaded value, so the wave goes on to its pivot chain while they travel -- with the padding select next to the loads, as in
// round 3, the wave sat out a full memory latency, about a microsecond, in front of every diagonal block);
// pivot_tile_finish, one step later, adds the children's entries (gather mode only) and puts the identity on the padding.
struct TileRaw {v4d a, b;};
__device__ __forceinline__ TileRaw pivot_tile_load(const FrontDesc & fd, const double * F, const double * B, bool clean, int row0, int col0, int lane)
{
  const int lr = lane & 15, lk = lane >> 4;
  const int row = row0 + lr, m = fd.m, ns = fd.ns;
  TileRaw t;
  t.b = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = col0 + lk + 4 * r;
    const bool want = row < ns && col <= row;
    const int64_t at = want ? row + (int64_t)col * m : 0;
    t.a[r] = F[at];
    if (B) {t.b[r] = B[at];}
    // self-cleaning fronts (round 6): what has been read is zero again for the next factorisation -- no 2 x 190 MB memset per
    // factorisation (L11 itself lives in LDS and leaves the kernel as W)
    if (clean && want) {
      const_cast<double *>(F)[at] = 0.0;
      if (B) {const_cast<double *>(B)[at] = 0.0;}
    }
  }
  return t;
}

__device__ __forceinline__ v4d pivot_tile_finish(const TileRaw & t, const SpaDev & d, const FrontDesc & fd, int nchild, int row0, int col0, int lane)
{
  v4d v = t.a + t.b;
  const int lr = lane & 15, lk = lane >> 4;
  const int row = row0 + lr, m = fd.m, ns = fd.ns, mp = m / 3;
  bool want[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = col0 + lk + 4 * r;
    want[r] = row < ns && col <= row;
  }
  for (int s = 0; s < nchild; ++s) {
    const ChildInfo c = child_info(d, fd, s);
    const int32_t * inv = d.cinv + fd.cinv_ptr + s * mp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = col0 + lk + 4 * r;
      const double g = gather_entry(d.fronts, c, inv, want[r] ? row : 0, want[r] ? col : 0);
      v[r] += want[r] ? g : 0.0;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int col = col0 + lk + 4 * r;
    v[r] = want[r] ? v[r] : ((row == col) ? 1.0 : 0.0);
  }
  return v;
}

// acc -= B A^T over the column blocks [p_lo, p_hi) of the LDS matrix X (row stride LD):  A = rows arow0.., B = rows brow0..
__device__ __forceinline__ void ll_accumulate(v4d & acc, const double * X, int LD, int arow0, int brow0, int p_lo, int p_hi, int lane)
{
  const int lr = lane & 15, lk = lane >> 4;
  const double * xa = X + (arow0 + lr) * LD + 4 * lk;
  const double * xb = X + (brow0 + lr) * LD + 4 * lk;
  // two column blocks at a time into two accumulators: a product is a chain of four DEPENDENT matrix-core operations (81 clocks
  // each, 64 of pipe), and a helper wave of k_potrf has up to six of them in a row per tile
  v4d other = v4d{0.0, 0.0, 0.0, 0.0};
  int p = p_lo;
  for (; p + 1 < p_hi; p += 2) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[NB * p + kk], xb[NB * p + kk], acc, 0, 0, 0);
      other = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[NB * (p + 1) + kk], xb[NB * (p + 1) + kk], other, 0, 0, 0);
    }
  }
  if (p < p_hi) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[NB * p + kk], xb[NB * p + kk], acc, 0, 0, 0);}
  }
  acc += other;
}

// Pivot block of every front of a level, LEFT-LOOKING, 512 threads (= 8 waves) per front.  Step jb (16 columns):
//   U  the tiles of column block jb get their final values: initial value (regular tiles: the front's entries + the
//      children's, requested one step ahead; identity rows: zero) minus the products with every earlier column block.
//      Wave 0 owns the DIAGONAL tile and its pivot chain (potrf_diag_mfma, with L_jj^-T riding along): the critical path of
//      the kernel.  Everything else belongs to the seven helper waves, dealt out task by task: the regular tiles (I, jb), the
//      identity-row tiles, block jb - 1 of W = L11^-T on its way to memory, the first term of the identity rows of block jb - 1
//      (the one that needs L_(jb-1)(jb-1)^-T, which only lives for one step) -- and the LOOK-AHEAD: the products of the NEXT
//      diagonal tile with the column blocks before jb, so that wave 0 opens step jb + 1 with one product instead of jb + 1.
//   R  row solves: every tile of the column times L_jj^-T (MFMA), one tile per wave.
// LDS holds only SOLVED tiles: L11 in the lower triangle, L11^-T strictly above it.
// Round 6: with four waves (round 3) every wave ran its tiles' products one after the other, each a chain of dependent
// v_mfma_f64_16x16x4 (81 clocks apiece, measured): on a 126-pivot front the three helper waves took 5-6 us per step, three
// times the pivot chain they were meant to hide behind (the kernel ran as long WITHOUT the chain).  The barriers of a step
// order LDS traffic only (lds_barrier): the next step's tile requests and the stores of W stay in flight across them.
constexpr int kPotrfThreads = 512, kPotrfHelpers = kPotrfThreads / 64 - 1;
__device__ __forceinline__ void potrf_body(const SpaDev & d, const FrontDesc & fd, int32_t * fail_flag, double * rhs, double * upd, int lds_nsp,
                                           long long * tbuf, int stamp_off)
{
  int tcount = 0;
  extern __shared__ double smem[];
  long long * stamps = reinterpret_cast<long long *>(smem) + stamp_off;
  PSTAMP();
  const int m = fd.m, ns = fd.ns;
  const int nsp = (ns + NB - 1) & ~(NB - 1), nt = nsp >> 4, LD = nsp + 2;
  const double * F = d.fronts + fd.off;
  double * W = d.winv + fd.woff;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double * X = smem;                                               // [nsp][LD]: lower = L11, strictly upper = L11^-T
  double * rd = smem + (size_t)lds_nsp * (lds_nsp + 2);            // [lds_nsp] reciprocal diagonal of L11 = diagonal of L11^-T
  double * yv = rd + lds_nsp;                                      // [lds_nsp] y1
  double * Xd = yv + lds_nsp;                                      // [2][16][XDS] L_jj^-T of the current / previous panel
  double * sc = Xd + 2 * NB * XDS;                                 // [64] scratch of the diagonal block's wave
  double * sb = sc + 64;                                           // [m] the front's slice of the right-hand side
  __shared__ int s_fail;
  if (tid == 0) {s_fail = 0;}
  const int nchild = fd.child_end - fd.child_ptr;
  const int nkids = front_nkids(d, fd);               // children whose update matrices are read in place
  const int mp = m / 3;
  const int h = wave - 1;                             // helper number
  const double * Bf = front_b(d, fd);
  const bool clean = d.scatter != 0;
  // Tiles requested from memory one step before they are used.  Helper h opens step jb with the regular tile (jb + 1 + h, jb)
  // (nt <= 8: a column block has at most seven tiles below the diagonal); the helper that will run the look-ahead of step jb
  // holds the diagonal tile (jb + 1, jb + 1); wave 0 loads the first diagonal tile and nothing after it.
  TileRaw cur, curd;
  auto look_helper = [&](int jb) {return (nt - jb - 1) % kPotrfHelpers;};
  auto prefetch = [&](int jb) {                        // for step jb
    if (jb >= nt) {return;}
    if (wave == 0) {
      if (jb == 0) {cur = pivot_tile_load(fd, F, Bf, clean, 0, 0, lane);}
      return;
    }
    const int I = jb + 1 + h;
    if (I < nt) {cur = pivot_tile_load(fd, F, Bf, clean, NB * I, NB * jb, lane);}
    if (jb + 1 < nt && look_helper(jb) == h) {curd = pivot_tile_load(fd, F, Bf, clean, NB * (jb + 1), NB * (jb + 1), lane);}
  };
  prefetch(0);
  // right-hand side: the pivots' entries + the children's forward-solve contributions (gathered through the same maps).  A
  // chain of two dependent loads per entry that nobody needs before the epilogue: the maps are requested here, the
  // contributions after the first step, the sums are formed behind the last one (entries beyond the first 512, and children
  // beyond the third, take the loop there).
  const int first = 3 * fd.first;
  const int tn = tid / 3;
  double rv = 0.0;
  int ca[3] = {-1, -1, -1};
  double cu[3] = {0.0, 0.0, 0.0};
  if (tid < m) {
    if (tid < ns) {rv = rhs[first + tid];}
#pragma unroll
    for (int s = 0; s < 3; ++s) {if (s < nchild) {ca[s] = d.cinv[fd.cinv_ptr + s * mp + tn];}}
  }
  PSTAMP();
  v4d res = v4d{0.0, 0.0, 0.0, 0.0};      // wave 0: the solved tile (jb, jb - 1), kept in registers from the row solves of step jb - 1
  for (int jb = 0; jb < nt; ++jb) {
    const int c0 = jb * NB;
    double * xd = Xd + (jb & 1) * NB * XDS;
    const double * xp = Xd + ((jb + 1) & 1) * NB * XDS;            // L^-T of the previous diagonal block
    if (wave == 0) {
      // the diagonal tile: everything but the last product was left in the tile's own place by the look-ahead of step jb - 1;
      // the last product's operand is `res` -- register r of the accumulator layout is column block r of the tile, which is what
      // a lane supplies to the matrix core for k = lk + 4 r
      v4d acc;
      if (jb == 0) {
        acc = pivot_tile_finish(cur, d, fd, nkids, 0, 0, lane);
      } else {
        const double * sp = X + (c0 + lr) * LD + c0 + lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {acc[r] = sp[4 * r];}
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-res[kk], res[kk], acc, 0, 0, 0);}
      }
      if (potrf_diag_mfma(acc, X + c0 * LD + c0, LD, lane, xd, rd + c0, sc)) {s_fail = 1;}
    } else {
      // regular tile (I, jb), I = jb + 1 + h
      {
        const int I = jb + 1 + h;
        if (I < nt) {
          v4d acc = pivot_tile_finish(cur, d, fd, nkids, NB * I, c0, lane);
          ll_accumulate(acc, X, LD, c0, NB * I, 0, jb, lane);
          double * cp = X + (NB * I + lr) * LD + c0 + lk;
#pragma unroll
          for (int r = 0; r < 4; ++r) {cp[4 * r] = acc[r];}
        }
      }
      // the other tasks of the step, dealt round robin starting behind the helpers that hold a regular tile
      int slot = look_helper(jb);
      auto mine = [&]() {const bool take = slot == h; slot = slot + 1 == kPotrfHelpers ? 0 : slot + 1; return take;};
      // look-ahead: the next diagonal tile's initial value - sum_{p < jb} L[jb+1][p] L[jb+1][p]^T, parked in the tile's own place
      if (jb + 1 < nt && mine()) {
        v4d acc = pivot_tile_finish(curd, d, fd, nkids, c0 + NB, c0 + NB, lane);
        ll_accumulate(acc, X, LD, c0 + NB, c0 + NB, 0, jb, lane);
        double * cp = X + (c0 + NB + lr) * LD + c0 + NB + lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {cp[4 * r] = acc[r];}
      }
      prefetch(jb + 1);
      // identity rows (I, jb), I < jb - 1: their value so far (block I's own term, written one step after block I) minus the
      // products with the column blocks I + 1 .. jb - 1 (longest first)
      for (int I = 0; I < jb - 1; ++I) {
        if (!mine()) {continue;}
        double * cp = X + (NB * I + lr) * LD + c0 + lk;
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {acc[r] = cp[4 * r];}
        ll_accumulate(acc, X, LD, c0, NB * I, I + 1, jb, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {cp[4 * r] = acc[r];}
      }
      if (jb > 0) {
        const int I = jb - 1;
        // block I's own term for this and the later column blocks J >= jb:  E[I][J] = -L_II^-T L[J][I]^T
        for (int J = jb; J < nt; ++J) {
          if (!mine()) {continue;}
          v4d acc = v4d{0.0, 0.0, 0.0, 0.0};
          const double * xa = X + (NB * J + lr) * LD + NB * I + 4 * lk;
          const double * xb = xp + lr * XDS + 4 * lk;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[kk], xb[kk], acc, 0, 0, 0);}
          double * cp = X + (NB * I + lr) * LD + NB * J + lk;
#pragma unroll
          for (int r = 0; r < 4; ++r) {cp[4 * r] = acc[r];}
        }
        // column block I of W is final: tiles (q <= I, I); W[(16 I + j) + (16 q + i) * nsp] = (L^-T)[16 q + i][16 I + j]
        for (int q = 0; q <= I; ++q) {
          if (!mine()) {continue;}
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int idx = lane + 64 * e, i = idx >> 4, j = idx & 15;
            W[(NB * I + j) + (int64_t)(NB * q + i) * nsp] = q == I ? xp[i * XDS + j] : X[(NB * q + i) * LD + NB * I + j];
          }
        }
      }
    }
    lds_barrier();                       // the tiles of column block jb and L_jj^-T are in place
    PSTAMP();
    if (jb == 0 && tid < m) {
      // second hop of the right-hand side's gather (the maps have long arrived)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (ca[s] >= 0) {cu[s] = upd[3 * (int64_t)fd.ch[s].rows_ptr + 3 * ca[s] + (tid - 3 * tn)];}
      }
    }
    // row solves, one 16-row tile per wave: rows below the block (L11 part) and the identity rows above it (the block's own
    // identity rows ARE xd).  Wave 0 takes the tile under the diagonal and keeps the result: its next diagonal tile needs it.
    if (wave == 0) {
      if (jb + 1 < nt) {res = potrf_rowsolve_tile(X + NB * (jb + 1) * LD + c0, LD, xd, lane);}
    } else {
      const int bi = h < jb ? h : h + 2;
      if (bi < nt) {(void)potrf_rowsolve_tile(X + NB * bi * LD + c0, LD, xd, lane);}
    }
    lds_barrier();
    PSTAMP();
  }
  // last column block of W
  {
    const int I = nt - 1;
    const double * xl = Xd + (I & 1) * NB * XDS;
    for (int q = wave; q <= I; q += kPotrfThreads / 64) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = lane + 64 * e, i = idx >> 4, j = idx & 15;
        W[(NB * I + j) + (int64_t)(NB * q + i) * nsp] = q == I ? xl[i * XDS + j] : X[(NB * q + i) * LD + NB * I + j];
      }
    }
  }
  // the right-hand side slice: the sums of the staged gather, then whatever it did not cover
  for (int t = tid; t < m; t += nthreads) {
    double v;
    int s0;
    if (t == tid) {
      v = (rv + cu[0]) + cu[1];
      v += cu[2];
      s0 = 3;
      if (nchild < 3) {v = rv; for (int s = 0; s < nchild; ++s) {v += cu[s];}}
    } else {
      v = t < ns ? rhs[first + t] : 0.0;
      s0 = 0;
    }
    const int tq = t / 3;
    for (int s = s0; s < nchild; ++s) {
      const ChildInfo c = child_info(d, fd, s);
      const int a = d.cinv[fd.cinv_ptr + s * mp + tq];
      const double u = upd[3 * (int64_t)c.rows_ptr + (a >= 0 ? 3 * a + (t - 3 * tq) : 0)];
      v += a >= 0 ? u : 0.0;
    }
    sb[t] = v;
  }
  __syncthreads();
  // y1 = L11^-1 b1 = W^T b1: four threads per entry, each with a quarter of the sum
  {
    const int j = tid >> 2, part = tid & 3;
    double acc = 0.0;
    if (j < ns) {
      // rows q < j of column j, split in four runs of equal length; the last run also takes the diagonal term
      const int q0 = (j * part) >> 2, q1 = (j * (part + 1)) >> 2;
#pragma unroll 8
      for (int q = q0; q < q1; ++q) {acc += X[q * LD + j] * sb[q];}
      if (part == 3) {acc += rd[j] * sb[j];}
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (j < ns && part == 0) {yv[j] = acc;}
  }
  __syncthreads();
  for (int t = tid; t < ns; t += nthreads) {rhs[first + t] = yv[t];}
  {
    double * uk = upd + 3 * (int64_t)fd.rows_ptr;
    for (int q = tid; q < m - ns; q += nthreads) {uk[q] = sb[ns + q];}
  }
  PSTAMP();
  if (tbuf && blockIdx.x == 0 && threadIdx.x == 0) {
    tbuf[0] = tcount; tbuf[63] = ((long long)m << 32) | ns;
    for (int i = 0; i < tcount; ++i) {tbuf[1 + i] = stamps[i];}
  }
  if (tid == 0 && s_fail) {atomicExch(fail_flag, 1);}
}

__global__ __launch_bounds__(kPotrfThreads) void k_potrf(SpaDev d, int first_front, int32_t * fail_flag, double * rhs, double * upd, int lds_nsp,
                                                         long long * tbuf, int stamp_off)
{
  const FrontDesc fd = d.desc[first_front + blockIdx.x];
  potrf_body(d, fd, fail_flag, rhs, upd, lds_nsp, tbuf, stamp_off);
}

// rows [prow0, prow0 + R) x columns [0, nsp) of the panel columns of a front -> LDS rows of stride LD (zeros outside
// nr rows / ns columns), sixteen loads in flight per thread; with kids, the children's update matrices are added (the
// same thread handles the same entry in every pass: no barrier in between)
template <int R>
__device__ __forceinline__ void stage_rows(double * S, int LD, const SpaDev & d, const FrontDesc & fd, int prow0, int nr, int ns, int nsp, int nkids,
                                           const double * Bfront, int tid, int nthreads)
{
  constexpr int LU = 16;
  const int m = fd.m, mp = m / 3;
  const double * Fcol0 = d.fronts + fd.off + prow0;
  const double * Bcol0 = Bfront ? Bfront + prow0 : nullptr;
  for (int base = 0; base < R * nsp; base += LU * nthreads) {
    double v[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int idx = base + u * nthreads + tid;
      const int c = idx / R, i = idx - c * R;
      const bool want = idx < R * nsp && i < nr && c < ns;
      v[u] = *(want ? Fcol0 + i + (int64_t)c * m : Fcol0);
    }
    if (Bcol0) {
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * nthreads + tid;
        const int c = idx / R, i = idx - c * R;
        const bool want = idx < R * nsp && i < nr && c < ns;
        const double * bp = want ? Bcol0 + i + (int64_t)c * m : Bcol0;
        v[u] += *bp;
        if (want) {*const_cast<double *>(bp) = 0.0;}          // self-cleaning (buffer A's entries become L21)
      }
    }
    for (int s = 0; s < nkids; ++s) {
      const ChildInfo ci = child_info(d, fd, s);
      const int32_t * inv = d.cinv + fd.cinv_ptr + s * mp;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * nthreads + tid;
        const int c = idx / R, i = idx - c * R;
        const bool want = idx < R * nsp && i < nr && c < ns;
        const double g = gather_entry(d.fronts, ci, inv, want ? prow0 + i : 0, want ? c : 0);
        v[u] += want ? g : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int idx = base + u * nthreads + tid;
      const int c = idx / R, i = idx - c * R;
      if (idx < R * nsp) {S[i * LD + c] = (i < nr && c < ns) ? v[u] : 0.0;}
    }
  }
}

// L21 = F21 W for a slab of R rows of one front (grid: front x slab), then upd -= L21 y1.  A wave owns column blocks
// of 16: the W entries of a block -- up to 128 k x 16 columns, 32 loads per lane -- are all requested before the first
// MFMA (a loop over k blocks with its loads inside is a chain of L2 latencies), and serve the R / 16 row tiles of the slab.
template <int R>
__global__ __launch_bounds__(512) void k_trsm(SpaDev d, int first_front, const double * __restrict__ rhs, double * upd, int lds_nsp)
{
  const FrontDesc fd = d.desc[first_front + blockIdx.x];
  const int m = fd.m, ns = fd.ns, nu = m - ns;
  const int r0 = R * (int)blockIdx.y;
  if (r0 >= nu) {return;}
  const int nr = min(R, nu - r0);
  const int nsp = (ns + NB - 1) & ~(NB - 1), nt = nsp >> 4, LD = nsp + 2;
  double * F = d.fronts + fd.off;
  const double * W = d.winv + fd.woff;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  extern __shared__ double smem[];
  double * S = smem;                                         // [R][LD] the slab of F21
  double * yv = smem + (size_t)R * (lds_nsp + 2);            // [lds_nsp] y1
  double * red = yv + lds_nsp;                               // [8][R] per-wave partial sums of L21 y1 (four or eight waves)
  const int first = 3 * fd.first;
  constexpr int RT = R / NB;
  constexpr int NTMAX = kPotrfMaxNs / NB;
  double * uk = upd + 3 * (int64_t)fd.rows_ptr;
  const double uold = tid < nr ? uk[r0 + tid] : 0.0;
  for (int j = tid; j < nsp; j += nthreads) {yv[j] = j < ns ? rhs[first + j] : 0.0;}
  for (int j = tid; j < 8 * R; j += nthreads) {red[j] = 0.0;}
  stage_rows<R>(S, LD, d, fd, ns + r0, nr, ns, nsp, front_nkids(d, fd), front_b(d, fd), tid, nthreads);
  __syncthreads();
  double part[RT];
#pragma unroll
  for (int it = 0; it < RT; ++it) {part[it] = 0.0;}
  for (int q = wave; q < nt; q += nwaves) {
    const int J = (q & 1) ? nt - 1 - (q >> 1) : (q >> 1);      // long and short K ranges alternate
    const double * wcol = W + (NB * J + lr);                   // (L^-T)[k][16 J + lr] = W[(16 J + lr) + k * nsp]
    double a[NTMAX][4];
#pragma unroll
    for (int K = 0; K < NTMAX; ++K) {
      const int Kc = K <= J ? K : 0;                           // always a valid address: no load under a branch
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {a[K][kk] = wcol[(int64_t)(NB * Kc + 4 * lk + kk) * nsp];}
    }
    v4d acc[RT];
#pragma unroll
    for (int it = 0; it < RT; ++it) {acc[it] = v4d{0.0, 0.0, 0.0, 0.0};}
#pragma unroll
    for (int K = 0; K < NTMAX; ++K) {
      if (K <= J) {                                            // wave-uniform
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int it = 0; it < RT; ++it) {
            acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[K][kk], S[(NB * it + lr) * LD + NB * K + 4 * lk + kk], acc[it], 0, 0, 0);
          }
        }
      }
    }
    // lane (lr, lk) holds L21[16 it + lr][16 J + lk + 4 r]
#pragma unroll
    for (int it = 0; it < RT; ++it) {
      const int row = NB * it + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = NB * J + lk + 4 * r;
        if (row < nr && col < ns) {F[(ns + r0 + row) + (int64_t)col * m] = acc[it][r];}
        part[it] += acc[it][r] * yv[col];                      // yv is zero on the padding columns
      }
    }
  }
  // forward solve: upd[row] -= sum_j L21[row][j] y1[j]; the four lane groups and the waves add up in a fixed order
#pragma unroll
  for (int it = 0; it < RT; ++it) {
    double p = part[it];
    p += __shfl_xor(p, 16);
    p += __shfl_xor(p, 32);
    if (lk == 0) {red[wave * R + NB * it + lr] = p;}
  }
  __syncthreads();
  if (tid < nr) {
    uk[r0 + tid] = uold - (((red[tid] + red[R + tid]) + (red[2 * R + tid] + red[3 * R + tid])) +
                           ((red[4 * R + tid] + red[5 * R + tid]) + (red[6 * R + tid] + red[7 * R + tid])));
  }
}

// F22 -= L21 L21^T, one TS x TS tile of the lower triangle per workgroup (grid: front x tile).  TS = 64: eight waves,
// each with a pair of 16 x 16 sub-tiles that share the row operand (two independent accumulators: the matrix pipe sees
// back-to-back MFMAs while the next operands arrive from LDS); TS = 32: four waves, one sub-tile each.
template <int TS>
__global__ __launch_bounds__(TS * 8) void k_syrk(SpaDev d, int first_front, int lds_nsp)
{
  const FrontDesc fd = d.desc[first_front + blockIdx.x];
  const int m = fd.m, ns = fd.ns, nu = m - ns;
  const int ntm = (nu + TS - 1) / TS;
  const int t = blockIdx.y;
  if (t >= ntm * (ntm + 1) / 2) {return;}
  int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  I -= (I * (I + 1) / 2 > t) ? 1 : 0;
  I += ((I + 1) * (I + 2) / 2 <= t) ? 1 : 0;
  const int J = t - I * (I + 1) / 2;
  const int nsp = (ns + NB - 1) & ~(NB - 1), LD = nsp + 2;
  double * F = d.fronts + fd.off;
  const double * Bf = front_b(d, fd);
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  extern __shared__ double smem[];
  double * XA = smem;                                               // rows of tile row I: [TS][LD]
  double * XB = (I == J) ? XA : smem + (size_t)TS * (lds_nsp + 2);  // rows of tile column J
  constexpr int UT = TS / 32;                  // sub-tiles per wave (side by side)
  constexpr int PW = (TS / NB) / UT;           // waves per sub-tile row
  const int si = wave / PW, sj0 = UT * (wave - si * PW);
  // accumulators first: their loads fly while the panel rows are staged
  v4d acc[UT];
  bool live[UT];
  const int row = TS * I + NB * si + lr;
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int sj = sj0 + u;
    live[u] = !(I == J && sj > si);
    const int col0 = TS * J + NB * sj + lk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = col0 + 4 * r;
      const bool ok = live[u] && row < nu && col <= row;
      const int64_t at = ok ? (ns + row) + (int64_t)(ns + col) * m : 0;
      double v = F[at];
      if (Bf) {v += Bf[at];}
      acc[u][r] = ok ? v : 0.0;
      if (d.scatter && ok) {                                   // self-cleaning: the tile leaves for the parent front (or is stored below)
        F[at] = 0.0;
        if (Bf) {const_cast<double *>(Bf)[at] = 0.0;}
      }
    }
  }
  // where the tile goes: into the parent front (round 6), or back where it came from
  const int mode = d.scatter ? (fd.flags & 3) : 0;
  int prow = 0, pcol[UT][4];
  int64_t pm = 0;
  double * P = nullptr;
  if (mode != 0) {
    const FrontDesc & pd = d.desc[fd.parent];
    const int32_t * rp = d.relpos + fd.relpos_ptr;
    const int rowc = min(row, nu - 1);
    prow = 3 * rp[rowc / 3] + rowc % 3;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int colc = min(TS * J + NB * (sj0 + u) + lk + 4 * r, nu - 1);
        pcol[u][r] = 3 * rp[colc / 3] + colc % 3;
      }
    }
    pm = pd.m;
    P = (mode == 1 ? d.fronts : d.fronts_b) + pd.off;
  }
  {
    // the children's update matrices join here (the front's own F22 holds only its assembled entries)
    const int nkids = front_nkids(d, fd), mp = m / 3;
    for (int s = 0; s < nkids; ++s) {
      const ChildInfo ci = child_info(d, fd, s);
      const int32_t * inv = d.cinv + fd.cinv_ptr + s * mp;
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        const int col0 = TS * J + NB * (sj0 + u) + lk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = col0 + 4 * r;
          const bool ok = live[u] && row < nu && col <= row;
          const double g = gather_entry(d.fronts, ci, inv, ok ? ns + row : 0, ok ? ns + col : 0);
          acc[u][r] += ok ? g : 0.0;
        }
      }
    }
  }
  stage_rows<TS>(XA, LD, d, fd, ns + TS * I, min(TS, nu - TS * I), ns, nsp, 0, nullptr, tid, nthreads);      // L21 is final: no children
  if (I != J) {stage_rows<TS>(XB, LD, d, fd, ns + TS * J, min(TS, nu - TS * J), ns, nsp, 0, nullptr, tid, nthreads);}
  __syncthreads();
  if (live[0]) {
    const double * xb = XA + (NB * si + lr) * LD + 4 * lk;
    const double * xa0 = XB + (NB * sj0 + lr) * LD + 4 * lk;
    if (UT == 2 && live[UT - 1]) {
      const double * xa1 = xa0 + NB * LD;
      for (int K = 0; K < nsp; K += NB) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double b = xb[K + kk];
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa0[K + kk], b, acc[0], 0, 0, 0);
          acc[UT - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa1[K + kk], b, acc[UT - 1], 0, 0, 0);
        }
      }
    } else {
      for (int K = 0; K < nsp; K += NB) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa0[K + kk], xb[K + kk], acc[0], 0, 0, 0);}
      }
    }
  }
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const int col0 = TS * J + NB * (sj0 + u) + lk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = col0 + 4 * r;
      if (live[u] && row < nu && col <= row) {
        if (mode != 0) {
          unsafeAtomicAdd(P + prow + pcol[u][r] * pm, acc[u][r]);
        } else {
          F[(ns + row) + (int64_t)(ns + col) * m] = acc[u][r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_front_update (round 6): k_trsm AND k_syrk of one front in ONE workgroup, L21 resident in LDS.
//
// On the wide levels of the tree (hundreds of fronts of 200-300 rows with 60-80 pivots) the split kernels are bound by what
// they re-read: every 64 x 64 tile of k_syrk stages its two 64-row slabs of L21 from the L2 again (level 0 of the 10k-node
// graph: 282 MB of slabs + 135 MB of update blocks + 67 MB of scatter in 61 us = the L2's rate, for 0.5 GFLOP), after k_trsm
// has written L21 and before that staged F21 once per slab.  A front of such a level fits one compute unit's LDS whole:
//   stage   F21 (both buffers, and the children read in place) -> S [nup][LD], W -> Wl [nsp][LDW], y1, the rows' places in the
//           parent
//   trsm    a wave per 16-row tile: L21 = F21 W with one accumulator per column block (independent MFMAs back to back),
//           written over the tile's own rows of S, to the front (the backward sweep reads it) and into upd -= L21 y1
//   syrk    16 x 16 sub-tiles of F22 -= L21 L21^T, two at a time per wave (they share the row operand), operands straight from
//           S; initial values from the front (both buffers), results added into the parent (or stored, Symbolic::scatter_mode)
// so F21, W and F22 are read once and L21 is written once.  The host sends the fronts of a level that fit (front_update_fits)
// here when there are enough of them to fill the chip; the larger ones (the head of the level: fronts are sorted by size) keep
// k_trsm / k_syrk.
// stage_rows with the number of rows known at run time (the whole F21 of a front at once, column by column)
__device__ __forceinline__ void stage_rows_rt(double * S, int LD, int R, const SpaDev & d, const FrontDesc & fd, int prow0, int nr, int ns, int nsp, int nkids,
                                              const double * Bfront, int tid, int nthreads)
{
  constexpr int LU = 8;
  const int m = fd.m, mp = m / 3;
  const double * Fcol0 = d.fronts + fd.off + prow0;
  const double * Bcol0 = Bfront ? Bfront + prow0 : nullptr;
  for (int base = 0; base < R * nsp; base += LU * nthreads) {
    double v[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int idx = base + u * nthreads + tid;
      const int c = idx / R, i = idx - c * R;
      const bool want = idx < R * nsp && i < nr && c < ns;
      v[u] = *(want ? Fcol0 + i + (int64_t)c * m : Fcol0);
    }
    if (Bcol0) {
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * nthreads + tid;
        const int c = idx / R, i = idx - c * R;
        const bool want = idx < R * nsp && i < nr && c < ns;
        const double * bp = want ? Bcol0 + i + (int64_t)c * m : Bcol0;
        v[u] += *bp;
        if (want) {*const_cast<double *>(bp) = 0.0;}          // self-cleaning (buffer A's entries become L21)
      }
    }
    for (int s = 0; s < nkids; ++s) {
      const ChildInfo ci = child_info(d, fd, s);
      const int32_t * inv = d.cinv + fd.cinv_ptr + s * mp;
#pragma unroll
      for (int u = 0; u < LU; ++u) {
        const int idx = base + u * nthreads + tid;
        const int c = idx / R, i = idx - c * R;
        const bool want = idx < R * nsp && i < nr && c < ns;
        const double g = gather_entry(d.fronts, ci, inv, want ? prow0 + i : 0, want ? c : 0);
        v[u] += want ? g : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int idx = base + u * nthreads + tid;
      const int c = idx / R, i = idx - c * R;
      if (idx < R * nsp) {S[(size_t)i * LD + c] = (i < nr && c < ns) ? v[u] : 0.0;}
    }
  }
}

constexpr int kFuThreads = 1024, kFuWaves = kFuThreads / 64;
constexpr int kFuMaxNt = 5;          // pivot blocks of up to 80 columns (the leaves of the dissection: 24 nodes = 72): one accumulator per column
                                     // block in the trsm phase, and 1024 threads leave a lane 128 registers

static size_t front_update_lds_bytes(int m, int ns)
{
  const int nu = m - ns, nup = (nu + NB - 1) & ~(NB - 1), nsp = (ns + NB - 1) & ~(NB - 1);
  return sizeof(double) * ((size_t)nup * (nsp + 2) + (size_t)nsp * (nsp + 4) + nsp + 8) + sizeof(int32_t) * (size_t)(nup / 3 + 8);
}
// LDS a front takes in k_front_update, 0 when it does not fit (or has no rows below the pivot block: nothing to do)
size_t spa_front_update_lds(int32_t m, int32_t ns)
{
  if (ns > NB * kFuMaxNt || m <= ns) {return 0;}
  const size_t b = front_update_lds_bytes(m, ns);
  return b <= 160 * 1024 - 512 ? b : 0;
}

template <int kWaves, int kMaxNt>
__device__ __forceinline__ void front_update_body(const SpaDev & d, const FrontDesc & fd, const double * rhs, double * upd)
{
  const int m = fd.m, ns = fd.ns, nu = m - ns;
  if (nu <= 0) {return;}
  const int nsp = (ns + NB - 1) & ~(NB - 1), nt = nsp >> 4, LD = nsp + 2, LDW = nsp + 4;
  const int nup = (nu + NB - 1) & ~(NB - 1), nrt = nup >> 4;
  double * F = d.fronts + fd.off;
  const double * Bf = front_b(d, fd);
  const double * W = d.winv + fd.woff;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  extern __shared__ double smem[];
  double * S = smem;                                   // [nup][LD]   F21, then L21
  double * Wl = S + (size_t)nup * LD;                  // [nsp][LDW]  Wl[k][j] = (L^-T)[k][j]
  double * yv = Wl + (size_t)nsp * LDW;                // [nsp]       y1
  int32_t * rpp = reinterpret_cast<int32_t *>(yv + nsp + 8);       // [nup / 3 + 1]  the struct rows' places in the parent (node units)
  const int first = 3 * fd.first;
  const int nkids = front_nkids(d, fd), mp = m / 3;
  const int mode = d.scatter ? (fd.flags & 3) : 0;
  // ---- stage ----
  for (int j = tid; j < nsp; j += nthreads) {yv[j] = j < ns ? rhs[first + j] : 0.0;}
  if (mode != 0) {
    const int32_t * rp = d.relpos + fd.relpos_ptr;
    for (int j = tid; j < nup / 3 + 1; j += nthreads) {rpp[j] = rp[min(j, nu / 3 - 1)];}
  }
  for (int idx = tid; idx < nsp * nsp; idx += nthreads) {
    const int k = idx / nsp, j = idx - k * nsp;
    Wl[k * LDW + j] = j >= (k & ~(NB - 1)) ? W[j + (int64_t)k * nsp] : 0.0;      // (left of the diagonal block W holds nothing)
  }
  stage_rows_rt(S, LD, nup, d, fd, ns, nu, ns, nsp, nkids, Bf, tid, nthreads);
  __syncthreads();
  // ---- trsm ----
  double * uk = upd + 3 * (int64_t)fd.rows_ptr;
  for (int it = wave; it < nrt; it += kWaves) {
    v4d acc[kMaxNt];
#pragma unroll
    for (int J = 0; J < kMaxNt; ++J) {acc[J] = v4d{0.0, 0.0, 0.0, 0.0};}
    const double * srow = S + (size_t)(NB * it + lr) * LD + 4 * lk;
#pragma unroll
    for (int K = 0; K < kMaxNt; ++K) {
      if (K < nt) {
        double b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {b[kk] = srow[NB * K + kk];}
        const double * wk = Wl + (size_t)(NB * K + 4 * lk) * LDW + lr;
#pragma unroll
        for (int J = K; J < kMaxNt; ++J) {
          if (J < nt) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(wk[kk * LDW + NB * J], b[kk], acc[J], 0, 0, 0);}
          }
        }
      }
    }
    // lane (lr, lk) holds L21[16 it + lr][16 J + lk + 4 r]: over the tile's own rows of S (nobody else reads them before the
    // barrier), to the front, and into the forward solve
    const int row = NB * it + lr;
    double part = 0.0;
#pragma unroll
    for (int J = 0; J < kMaxNt; ++J) {
      if (J < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = NB * J + lk + 4 * r;
          S[(size_t)row * LD + col] = acc[J][r];
          if (row < nu && col < ns) {F[(ns + row) + (int64_t)col * m] = acc[J][r];}
          part += acc[J][r] * yv[col];                             // yv is zero on the padding columns
        }
      }
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    if (lk == 0 && row < nu) {uk[row] -= part;}
  }
  __syncthreads();
  // ---- syrk ----
  const FrontDesc * pd = mode != 0 ? d.desc + fd.parent : nullptr;
  const int64_t pm = pd ? pd->m : 0;
  double * P = pd ? (mode == 1 ? d.fronts : d.fronts_b) + pd->off : nullptr;
  // units: (si, pair of sub-tile columns 2 u, 2 u + 1), si = 0 .. nrt - 1, u = 0 .. si / 2, numbered row by row
  int unit = wave;
  int si = 0, ubase = 0;                               // ubase = units before tile row si
  const int nunits = [&]() {int n = 0; for (int q = 0; q < nrt; ++q) {n += q / 2 + 1;} return n;}();
  for (; unit < nunits; unit += kWaves) {
    while (unit >= ubase + si / 2 + 1) {ubase += si / 2 + 1; ++si;}
    const int sj0 = 2 * (unit - ubase);
    const bool two = sj0 + 1 <= si;
    const int row = NB * si + lr;
    v4d acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = NB * (sj0 + u) + lk + 4 * r;
        const bool ok = (u == 0 || two) && row < nu && col <= row;
        const int64_t at = ok ? (ns + row) + (int64_t)(ns + col) * m : 0;
        double v = F[at];
        if (Bf) {v += Bf[at];}
        acc[u][r] = ok ? v : 0.0;
        if (d.scatter && ok) {
          F[at] = 0.0;
          if (Bf) {const_cast<double *>(Bf)[at] = 0.0;}
        }
      }
    }
    for (int s = 0; s < nkids; ++s) {
      const ChildInfo ci = child_info(d, fd, s);
      const int32_t * inv = d.cinv + fd.cinv_ptr + s * mp;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = NB * (sj0 + u) + lk + 4 * r;
          const bool ok = (u == 0 || two) && row < nu && col <= row;
          const double g = gather_entry(d.fronts, ci, inv, ok ? ns + row : 0, ok ? ns + col : 0);
          acc[u][r] += ok ? g : 0.0;
        }
      }
    }
    const double * xb = S + (size_t)(NB * si + lr) * LD + 4 * lk;
    const double * xa0 = S + (size_t)(NB * sj0 + lr) * LD + 4 * lk;
    const double * xa1 = xa0 + (two ? NB * LD : 0);
    if (two) {
      for (int K = 0; K < nsp; K += NB) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const double b = xb[K + kk];
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa0[K + kk], b, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa1[K + kk], b, acc[1], 0, 0, 0);
        }
      }
    } else {
      for (int K = 0; K < nsp; K += NB) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa0[K + kk], xb[K + kk], acc[0], 0, 0, 0);}
      }
    }
    const int rowc = min(row, nu - 1);
    const int64_t prow = mode != 0 ? 3 * (int64_t)rpp[rowc / 3] + rowc % 3 : 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = NB * (sj0 + u) + lk + 4 * r;
        if ((u == 0 || two) && row < nu && col <= row) {
          if (mode != 0) {
            const int64_t pcol = 3 * (int64_t)rpp[col / 3] + col % 3;
            unsafeAtomicAdd(P + prow + pcol * pm, acc[u][r]);
          } else {
            F[(ns + row) + (int64_t)(ns + col) * m] = acc[u][r];
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(kFuThreads) void k_front_update(SpaDev d, int first_front, const double * __restrict__ rhs, double * upd)
{
  const FrontDesc fd = d.desc[first_front + blockIdx.x];
  front_update_body<kFuWaves, kFuMaxNt>(d, fd, rhs, upd);
}

void spa_launch_front_update(const SpaDev & d, int32_t first_front, int32_t n, size_t lds_bytes, const double * rhs, double * upd, void * stream)
{
  if (n <= 0) {return;}
  static std::atomic<unsigned long long> done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(k_front_update), 160 * 1024 - 256, done);
  hipLaunchKernelGGL(k_front_update, dim3(n), dim3(kFuThreads), lds_bytes, (hipStream_t)stream, d, first_front, rhs, upd);
}

// Backward solve of a level with W = L11^-T:  x1 = W (y1 - L21^T x2), two sets of independent dot products, one wave per
// pair of columns.  (A variant that requested every entry of L21 and W a wave needs right after the descriptor -- 80 loads
// per lane, fully unrolled -- was SLOWER, 306 us per sweep against 177: these kernels run once per level, and a long
// straight-line body is paid for in instruction fetch.)
template <bool kNarrow>
__global__ __launch_bounds__(1024) void k_backward3(SpaDev d, int first_front, double * rhs)
{
  const FrontDesc fd = d.desc[first_front + blockIdx.x];
  const int m = fd.m, ns = fd.ns, nu = m - ns;
  const int nsp = (ns + NB - 1) & ~(NB - 1);
  const double * F = d.fronts + fd.off;
  const double * W = d.winv + fd.woff;
  const int first = 3 * fd.first;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nwaves = nthreads >> 6;
  extern __shared__ double sb[];                 // [m] : w (pivots) | x2 (struct rows), then [nsp] x1
  double * xo = sb + m;
  const int32_t * rows = d.front_rows + fd.rows_ptr;
  for (int t = tid; t < ns; t += nthreads) {sb[t] = rhs[first + t];}
  for (int q = tid; q < nu; q += nthreads) {sb[ns + q] = rhs[3 * rows[q / 3] + q % 3];}
  __syncthreads();
  if (kNarrow) {
    // NARROW levels (a handful of fronts: latency is everything).  Eight lanes per column, 128 columns at a time: every column
    // of the pivot block is in flight at once (with a wave per pair of columns a 126-pivot front took its 63 pairs in four
    // rounds of sixteen, each round a memory latency of its own).  On the wide levels the 64-byte pieces of this form cost more
    // than the rounds (measured: 17.6 -> 21.0 us on the 69-front level), so they keep the wave per pair.
    const int grp = tid >> 3, sub = tid & 7, ngrp = nthreads >> 3;
    // w[c] = y1[c] - L21[:, c] . x2
    for (int c = grp; c < ns; c += ngrp) {
      const double * col = F + ns + (int64_t)c * m;
      double a0 = 0.0;
#pragma unroll 4
      for (int i = sub; i < nu; i += 8) {
        a0 += col[i] * sb[ns + i];
        if (d.scatter) {const_cast<double *>(col)[i] = 0.0;}        // self-cleaning fronts: this was the last reader of L21
      }
      a0 += __shfl_xor(a0, 1); a0 += __shfl_xor(a0, 2); a0 += __shfl_xor(a0, 4);
      if (sub == 0) {sb[c] -= a0;}
    }
    __syncthreads();
    // x1[c] = sum_{j >= c} (L^-T)[c][j] w[j],  (L^-T)[c][j] = W[j + c * nsp]  (zeros left of the diagonal)
    for (int c = grp; c < ns; c += ngrp) {
      const double * w0 = W + (int64_t)c * nsp;
      double a0 = 0.0;
#pragma unroll 4
      for (int j = c + sub; j < ns; j += 8) {a0 += w0[j] * sb[j];}
      a0 += __shfl_xor(a0, 1); a0 += __shfl_xor(a0, 2); a0 += __shfl_xor(a0, 4);
      if (sub == 0) {xo[c] = a0;}
    }
  } else {
    // w[c] = y1[c] - L21[:, c] . x2
    for (int c = 2 * wave; c < ns; c += 2 * nwaves) {
      const double * col0 = F + ns + (int64_t)c * m;
      const double * col1 = col0 + (c + 1 < ns ? m : 0);
      double a0 = 0.0, a1 = 0.0;
      for (int i = lane; i < nu; i += 64) {
        const double x = sb[ns + i];
        a0 += col0[i] * x;
        a1 += col1[i] * x;
        if (d.scatter) {                        // self-cleaning fronts: this was the last reader of L21
          const_cast<double *>(col0)[i] = 0.0;
          if (c + 1 < ns) {const_cast<double *>(col1)[i] = 0.0;}
        }
      }
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {a0 += __shfl_xor(a0, s); a1 += __shfl_xor(a1, s);}
      if (lane == 0) {sb[c] -= a0; if (c + 1 < ns) {sb[c + 1] -= a1;}}
    }
    __syncthreads();
    // x1[c] = sum_{j >= c} (L^-T)[c][j] w[j],  (L^-T)[c][j] = W[j + c * nsp]  (zeros left of the diagonal)
    for (int c = 2 * wave; c < ns; c += 2 * nwaves) {
      const double * w0 = W + (int64_t)c * nsp;
      const double * w1 = w0 + (c + 1 < ns ? nsp : 0);
      double a0 = 0.0, a1 = 0.0;
      for (int j = c + lane; j < ns; j += 64) {
        const double wj = sb[j];
        a0 += w0[j] * wj;
        a1 += j > c ? w1[j] * wj : 0.0;       // row c + 1 starts at column c + 1: left of it W holds nothing (not even zeros when
      }                                        // c + 1 opens a new block of 16)
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) {a0 += __shfl_xor(a0, s); a1 += __shfl_xor(a1, s);}
      if (lane == 0) {xo[c] = a0; if (c + 1 < ns) {xo[c + 1] = a1;}}
    }
  }
  __syncthreads();
  for (int t = tid; t < ns; t += nthreads) {rhs[first + t] = xo[t];}
}

// self-cleaning fronts: the update matrices that stayed in place (children their parent reads through cinv) are zeroed here,
// after the factorisation; list = those fronts, one workgroup per (front, 16 columns)
__global__ __launch_bounds__(256) void k_zero_update_blocks(SpaDev d, const int32_t * __restrict__ list)
{
  const FrontDesc fd = d.desc[list[blockIdx.x]];
  const int m = fd.m, ns = fd.ns, nu = m - ns;
  double * U = d.fronts + fd.off + ns + (int64_t)ns * m;
  for (int c = 16 * (int)blockIdx.y + ((int)threadIdx.x >> 4); c < nu; c += 16 * (int)gridDim.y) {
    for (int i = c + ((int)threadIdx.x & 15); i < nu; i += 16) {U[i + (int64_t)c * m] = 0.0;}
  }
}
void spa_launch_zero_update_blocks(const SpaDev & d, const int32_t * list, int32_t n, int32_t max_m, void * stream)
{
  if (n <= 0) {return;}
  hipLaunchKernelGGL(k_zero_update_blocks, dim3(n, std::max(1, std::min(16, max_m / 16))), dim3(256), 0, (hipStream_t)stream, d, list);
}

// debugging aid (kh_spa_set_debug bit 0): entries of a buffer that are not zero (bit pattern), counted into *count
__global__ __launch_bounds__(256) void k_count_nonzero(const double * __restrict__ p, int64_t n, int32_t * count)
{
  int mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    mine += __double_as_longlong(p[i]) != 0 ? 1 : 0;
  }
  if (mine) {atomicAdd(count, mine);}
}
void spa_launch_count_nonzero(const double * p, int64_t n, int32_t * count, void * stream)
{
  hipLaunchKernelGGL(k_count_nonzero, dim3(1024), dim3(256), 0, (hipStream_t)stream, p, n, count);
}

static size_t potrf_lds_bytes(int nsp, int max_m)
{
  return sizeof(double) * ((size_t)nsp * (nsp + 2) + 2 * (size_t)nsp + 2 * NB * XDS + 64 + (size_t)max_m + 8);
}

bool spa_level_pipeline_fits(int32_t max_m, int32_t max_ns)
{
  const int nsp = (max_ns + NB - 1) & ~(NB - 1);
  return max_ns <= kPotrfMaxNs && potrf_lds_bytes(nsp, max_m) <= 160 * 1024 - 256 && sizeof(double) * ((size_t)max_m + nsp + 8) <= 64 * 1024;
}

static void pipeline_attributes()
{
  // (per device: the attribute is kept per device, and solvers of several devices may live in one process)
  const int big = 160 * 1024 - 256;       // static LDS (a flag word) counts against the same 160 KB
  static std::atomic<unsigned long long> done[5] = {{0}, {0}, {0}, {0}, {0}};
  allow_dynamic_lds(reinterpret_cast<const void *>(k_potrf), big, done[0]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_trsm<32>), big, done[1]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_trsm<64>), big, done[2]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_syrk<32>), big, done[3]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_syrk<64>), big, done[4]);
}

void spa_launch_potrf_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, int32_t * fail_flag,
                            double * rhs, double * upd, void * stream)
{
  if (n <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  const int nsp = (max_ns + NB - 1) & ~(NB - 1);
  pipeline_attributes();
  static long long * tbuf = nullptr;
  static const bool timing = std::getenv("KH_SPA_TIMING") != nullptr;
  if (timing && !tbuf) {(void)hipHostMalloc(reinterpret_cast<void **>(&tbuf), 128 * sizeof(long long), hipHostMallocDefault);}
  const size_t lds = potrf_lds_bytes(nsp, max_m);
  hipLaunchKernelGGL(k_potrf, dim3(n), dim3(kPotrfThreads), lds + (timing ? 512 : 0), s, d, first_front, fail_flag, rhs, upd, nsp,
                     timing ? tbuf : (long long *)nullptr, (int)(lds / 8));
  if (timing) {
    (void)hipStreamSynchronize(s);
    std::fprintf(stderr, "[k_potrf] n=%d front0 m=%lld ns=%lld stamps (shader clocks):", n, tbuf[63] >> 32, tbuf[63] & 0xffffffff);
    for (int i = 1; i < (int)tbuf[0]; ++i) {std::fprintf(stderr, " %lld", tbuf[1 + i] - tbuf[i]);}
    std::fprintf(stderr, "\n");
  }
}

void spa_launch_update_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, double * rhs, double * upd, void * stream)
{
  if (n <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  const int nsp = (max_ns + NB - 1) & ~(NB - 1);
  pipeline_attributes();
  const int max_nu = max_m - 3;       // a front has at least one pivot node; an upper bound is enough for the grid
  if (max_nu <= 0) {return;}
  // slabs / tiles: small enough that a narrow level still spreads over the chip, large enough that a wide one does not
  // drown in workgroups
  static const int force_r = std::getenv("KH_SPA_TRSM_R") ? std::atoi(std::getenv("KH_SPA_TRSM_R")) : 0;
  static const int force_ts = std::getenv("KH_SPA_SYRK_TS") ? std::atoi(std::getenv("KH_SPA_SYRK_TS")) : 0;
  const int slabs32 = (max_nu + 31) / 32;
  const bool r64 = force_r ? force_r == 64 : (int64_t)n * slabs32 >= 1024;
  if (r64) {
    hipLaunchKernelGGL(k_trsm<64>, dim3(n, (max_nu + 63) / 64), dim3(256), sizeof(double) * ((size_t)64 * (nsp + 2) + nsp + 8 * 64 + 8), s, d, first_front, rhs, upd, nsp);
  } else {
    // a narrow level: eight waves, so that every column block of the pivots has a wave of its own (one round of W loads, not two)
    const int threads = (int64_t)n * slabs32 <= 256 ? 512 : 256;
    hipLaunchKernelGGL(k_trsm<32>, dim3(n, slabs32), dim3(threads), sizeof(double) * ((size_t)32 * (nsp + 2) + nsp + 8 * 32 + 8), s, d, first_front, rhs, upd, nsp);
  }
  const int nt32 = (max_nu + 31) / 32, nt64 = (max_nu + 63) / 64;
  const bool t64 = force_ts ? force_ts == 64 : (int64_t)n * (nt32 * (nt32 + 1) / 2) >= 2048;
  if (t64) {
    hipLaunchKernelGGL(k_syrk<64>, dim3(n, nt64 * (nt64 + 1) / 2), dim3(512), sizeof(double) * ((size_t)2 * 64 * (nsp + 2) + 8), s, d, first_front, nsp);
  } else {
    hipLaunchKernelGGL(k_syrk<32>, dim3(n, nt32 * (nt32 + 1) / 2), dim3(256), sizeof(double) * ((size_t)2 * 32 * (nsp + 2) + 8), s, d, first_front, nsp);
  }
}

void spa_launch_backward3_level(const SpaDev & d, int32_t first_front, int32_t n, int32_t max_m, int32_t max_ns, double * rhs, void * stream)
{
  if (n <= 0) {return;}
  const int nsp = (max_ns + NB - 1) & ~(NB - 1);
  if (n <= 12) {
    hipLaunchKernelGGL(k_backward3<true>, dim3(n), dim3(1024), sizeof(double) * ((size_t)max_m + nsp + 8), (hipStream_t)stream, d, first_front, rhs);
  } else {
    hipLaunchKernelGGL(k_backward3<false>, dim3(n), dim3(max_ns <= 48 ? 256 : 1024), sizeof(double) * ((size_t)max_m + nsp + 8), (hipStream_t)stream, d, first_front, rhs);
  }
}

}  // namespace kh
