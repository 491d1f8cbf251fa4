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
#include <cstdint>
#include "spa_internal.hpp"

#pragma clang fp contract(fast)

namespace kh
{

constexpr double kPiD = 3.14159265358979323846;
constexpr double kTwoPiD = 2.0 * kPiD;

// ceres_utils.h:27-32
__device__ __forceinline__ double d_normalize_angle(double a) {return a - kTwoPiD * floor((a + kPiD) / kTwoPiD);}

// ---------------------------------------------------------------------------------------------
template <bool kJac>
__global__ __launch_bounds__(256) void k_edge_lin(SpaDev d, const double * __restrict__ x)
{
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d.n_edges) {return;}
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
  const double f0 = U[0] * r0 + U[1] * r1 + U[2] * r2;
  const double f1 = U[4] * r1 + U[5] * r2;
  const double f2 = U[8] * r2;
  d.edge_cost[e] = f0 * f0 + f1 * f1 + f2 * f2;
  if (!kJac) {return;}
  double * out = d.edge_lin + 21 * (size_t)e;
  out[0] = f0; out[1] = f1; out[2] = f2;
  // raw Jacobians (rows = residual, cols = xa, ya, ta | xb, yb, tb)
  const double ja[9] = {-c, -s, -s * dx + c * dy, s, -c, -c * dx - s * dy, 0.0, 0.0, -1.0};
  const double jb[9] = {c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    out[3 + 0 + col] = U[0] * ja[col] + U[1] * ja[3 + col] + U[2] * ja[6 + col];
    out[3 + 3 + col] = U[4] * ja[3 + col] + U[5] * ja[6 + col];
    out[3 + 6 + col] = U[8] * ja[6 + col];
    out[12 + 0 + col] = U[0] * jb[col] + U[1] * jb[3 + col] + U[2] * jb[6 + col];
    out[12 + 3 + col] = U[4] * jb[3 + col] + U[5] * jb[6 + col];
    out[12 + 6 + col] = U[8] * jb[6 + col];
  }
}

__global__ __launch_bounds__(256) void k_gather_H(SpaDev d)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
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

void spa_launch_linearize(const SpaDev & d, const double * x, double * cost_out, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  if (d.n_edges > 0) {
    hipLaunchKernelGGL(k_edge_lin<true>, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d, x);
    hipLaunchKernelGGL(k_gather_H, dim3((d.n_slots * 9 + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_gather_g, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, s, d);
  }
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, d.edge_cost, d.n_edges, 0.5, cost_out);
}

void spa_launch_cost(const SpaDev & d, const double * x, double * cost_out, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  if (d.n_edges > 0) {
    hipLaunchKernelGGL(k_edge_lin<false>, dim3((d.n_edges + 255) / 256), dim3(256), 0, s, d, x);
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

__global__ void k_diag(SpaDev d, const double * scale, double * diag, double lo, double hi)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_free * 3) {return;}
  const int i = t / 3, c = t - i * 3;
  double v = scale[t] * d.H[(size_t)d.bsr_diag_slot[i] * 9 + c * 3 + c] * scale[t];
  v = v < lo ? lo : v;          // levenberg_marquardt_strategy.cc: clamp to [min_diagonal, max_diagonal]
  v = v > hi ? hi : v;
  diag[t] = v;
}
void spa_launch_diag(const SpaDev & d, const double * scale, double * diag_out, double min_diag, double max_diag, void * stream)
{
  hipLaunchKernelGGL(k_diag, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, scale, diag_out, min_diag, max_diag);
}

__global__ void k_assemble(SpaDev d, const int32_t * slot_row, const double * scale, const double * diagonal, double inv_radius)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_slots * 9) {return;}
  const int slot = t / 9, el = t - slot * 9;
  const int64_t dest = d.slot_dest[slot];
  if (dest < 0) {return;}
  const int r = el / 3, c = el - r * 3;
  const int i = slot_row[slot], j = d.bsr_col[slot];
  double v = scale[3 * i + r] * d.H[t] * scale[3 * j + c];
  if (i == j && r == c) {v += diagonal[3 * i + r] * inv_radius;}
  d.fronts[dest + r + (int64_t)c * d.slot_ld[slot]] = v;
}

// slot_row is stored right behind bsr_col by the host (bsr_col + n_slots)
void spa_launch_assemble(const SpaDev & d, const double * scale, const double * diagonal, double inv_radius, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(d.fronts, 0, sizeof(double) * d.fronts_size, s);
  hipLaunchKernelGGL(k_assemble, dim3((d.n_slots * 9 + 255) / 256), dim3(256), 0, s, d, d.bsr_col + d.n_slots, scale, diagonal, inv_radius);
}

__global__ void k_make_rhs(SpaDev d, const double * scale, double * rhs)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_free * 3) {return;}
  const int i = t / 3, c = t - i * 3;
  rhs[3 * d.elim_of_free[i] + c] = scale[t] * d.g[t];
}
void spa_launch_make_rhs(const SpaDev & d, const double * scale, double * rhs, void * stream)
{
  hipLaunchKernelGGL(k_make_rhs, dim3((d.n_free * 3 + 255) / 256), dim3(256), 0, (hipStream_t)stream, d, scale, rhs);
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
// K6b: multifrontal partial Cholesky, one workgroup per front.
constexpr int NB = 16;

__global__ __launch_bounds__(256) void k_factor(SpaDev d, const int32_t * __restrict__ level_fronts, int32_t * fail_flag)
{
  const int k = level_fronts[blockIdx.x];
  const int m = d.front_m[k], ns = d.front_ns[k];
  double * F = d.fronts + d.front_off[k];
  const int tid = threadIdx.x;

  // 1. extend-add the children's update matrices (lower triangles)
  for (int ci = d.child_ptr[k]; ci < d.child_ptr[k + 1]; ++ci) {
    const int c = d.child_list[ci];
    const int mc = d.front_m[c], nsc = d.front_ns[c], nuc = mc - nsc;
    const double * Uc = d.fronts + d.front_off[c] + nsc + (int64_t)nsc * mc;
    const int32_t * rp = d.relpos + d.relpos_ptr[c];
    for (int idx = tid; idx < nuc * nuc; idx += 256) {
      const int a = idx % nuc, b = idx / nuc;
      if (a < b) {continue;}
      const int pa = 3 * rp[a / 3] + a % 3, pb = 3 * rp[b / 3] + b % 3;
      F[pa + (int64_t)pb * m] += Uc[a + (int64_t)b * mc];
    }
    __syncthreads();
  }

  // 2. blocked right-looking partial Cholesky of the first ns columns
  __shared__ double Ld[NB][NB + 1];
  __shared__ int s_fail;
  if (tid == 0) {s_fail = 0;}
  __syncthreads();
  for (int jb = 0; jb < ns; jb += NB) {
    const int nb = min(NB, ns - jb);
    // (a) diagonal block: wave 0, lane l owns row l of the nb x nb block in registers
    if (tid < 64) {
      double row[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        row[c] = (tid < nb && c < nb && c <= tid) ? F[(jb + tid) + (int64_t)(jb + c) * m] : ((c == tid) ? 1.0 : 0.0);
      }
      bool bad = false;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        double djj = __shfl(row[j], j);
        if (!(djj > 0.0)) {bad = true; djj = 1.0;}
        const double inv = 1.0 / sqrt(djj);
        if (tid == j) {row[j] = sqrt(djj);} else if (tid > j) {row[j] *= inv;}
#pragma unroll
        for (int c = j + 1; c < NB; ++c) {
          const double lcj = __shfl(row[j], c);       // L[c][j]
          if (tid >= c) {row[c] -= row[j] * lcj;}
        }
      }
      if (tid < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) {Ld[tid][c] = row[c];}
        if (tid < nb) {
          for (int c = 0; c <= tid; ++c) {F[(jb + tid) + (int64_t)(jb + c) * m] = row[c];}
        }
      }
      if (bad && tid < nb) {s_fail = 1;}
    }
    __syncthreads();
    // (b) panel: X = F[rows, jb:jb+nb] * Ld^{-T}, rows below the diagonal block
    const int r0 = jb + nb;
    for (int i = r0 + tid; i < m; i += 256) {
      double xr[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {xr[c] = (c < nb) ? F[i + (int64_t)(jb + c) * m] : 0.0;}
#pragma unroll
      for (int c = 0; c < NB; ++c) {
        if (c < nb) {
          double v = xr[c];
#pragma unroll
          for (int q = 0; q < c; ++q) {v -= xr[q] * Ld[c][q];}
          xr[c] = v / Ld[c][c];
        }
      }
#pragma unroll
      for (int c = 0; c < NB; ++c) {if (c < nb) {F[i + (int64_t)(jb + c) * m] = xr[c];}}
    }
    __syncthreads();
    // (c) trailing update (lower triangle): F[i][j] -= X[i,:] . X[j,:]   for r0 <= j <= i < m
    for (int i = r0 + tid; i < m; i += 256) {
      double xi[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) {xi[c] = (c < nb) ? F[i + (int64_t)(jb + c) * m] : 0.0;}
      for (int j = r0; j <= i; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < NB; ++c) {acc += xi[c] * ((c < nb) ? F[j + (int64_t)(jb + c) * m] : 0.0);}
        F[i + (int64_t)j * m] -= acc;
      }
    }
    __syncthreads();
  }
  if (tid == 0 && s_fail) {atomicExch(fail_flag, 1);}
}

void spa_launch_factor_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, int32_t * fail_flag, void * stream)
{
  if (n <= 0) {return;}
  hipLaunchKernelGGL(k_factor, dim3(n), dim3(256), 0, (hipStream_t)stream, d, level_fronts, fail_flag);
}

// ---------------------------------------------------------------------------------------------
// K6c: triangular solves, one 64-lane workgroup per front.  rhs is in elimination order.
constexpr int kMaxNs = 3072;    // pivot columns of the largest supported front (host checks)

__global__ __launch_bounds__(64) void k_forward(SpaDev d, const int32_t * __restrict__ level_fronts, double * rhs)
{
  const int k = level_fronts[blockIdx.x];
  const int m = d.front_m[k], ns = d.front_ns[k], nu = m - ns;
  const double * F = d.fronts + d.front_off[k];
  const int first = 3 * d.front_first[k];
  const int tid = threadIdx.x;
  extern __shared__ double sb[];
  for (int t = tid; t < ns; t += 64) {sb[t] = rhs[first + t];}
  __syncthreads();
  for (int j = 0; j < ns; ++j) {
    const double yj = sb[j] / F[j + (int64_t)j * m];
    __syncthreads();
    if (tid == 0) {sb[j] = yj;}
    for (int i = j + 1 + tid; i < ns; i += 64) {sb[i] -= F[i + (int64_t)j * m] * yj;}
    __syncthreads();
  }
  for (int t = tid; t < ns; t += 64) {rhs[first + t] = sb[t];}
  const int32_t * rows = d.front_rows + d.front_rows_ptr[k];
  for (int q = tid; q < nu; q += 64) {
    double acc = 0.0;
    for (int t = 0; t < ns; ++t) {acc += F[(ns + q) + (int64_t)t * m] * sb[t];}
    atomicAdd(&rhs[3 * rows[q / 3] + q % 3], -acc);
  }
}

__global__ __launch_bounds__(64) void k_backward(SpaDev d, const int32_t * __restrict__ level_fronts, double * rhs)
{
  const int k = level_fronts[blockIdx.x];
  const int m = d.front_m[k], ns = d.front_ns[k], nu = m - ns;
  const double * F = d.fronts + d.front_off[k];
  const int first = 3 * d.front_first[k];
  const int tid = threadIdx.x;
  extern __shared__ double sb[];
  const int32_t * rows = d.front_rows + d.front_rows_ptr[k];
  // w = y - L21^T x_struct
  for (int t = tid; t < ns; t += 64) {
    double acc = rhs[first + t];
    const double * col = F + ns + (int64_t)t * m;
    for (int q = 0; q < nu; ++q) {acc -= col[q] * rhs[3 * rows[q / 3] + q % 3];}
    sb[t] = acc;
  }
  __syncthreads();
  for (int j = ns - 1; j >= 0; --j) {
    const double xj = sb[j] / F[j + (int64_t)j * m];
    __syncthreads();
    if (tid == 0) {sb[j] = xj;}
    for (int i = tid; i < j; i += 64) {sb[i] -= F[j + (int64_t)i * m] * xj;}
    __syncthreads();
  }
  for (int t = tid; t < ns; t += 64) {rhs[first + t] = sb[t];}
}

void spa_launch_forward_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, double * rhs, void * stream)
{
  if (n <= 0) {return;}
  hipLaunchKernelGGL(k_forward, dim3(n), dim3(64), sizeof(double) * kMaxNs, (hipStream_t)stream, d, level_fronts, rhs);
}
void spa_launch_backward_level(const SpaDev & d, const int32_t * level_fronts, int32_t n, double * rhs, void * stream)
{
  if (n <= 0) {return;}
  hipLaunchKernelGGL(k_backward, dim3(n), dim3(64), sizeof(double) * kMaxNs, (hipStream_t)stream, d, level_fronts, rhs);
}

}  // namespace kh
