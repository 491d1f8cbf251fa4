// Microbenchmark of the 16 x 16 diagonal-block factorisations of k_potrf (one wave, LDS in and out):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/diag_probe.hip -o /tmp/diag_probe && /tmp/diag_probe
// prints clocks per call of the round-3 pivot chain (potrf_diag) and of the matrix-core form (potrf_diag_mfma), and the
// error of both against a host Cholesky.
#ifdef STAMPS
#define KH_DIAG_STAMPS
#endif
#include "../slam_toolbox_amd/csrc/spa_kernels.hip"
#include <cmath>
#include <vector>
using namespace kh;

template <int kMode>
__global__ __launch_bounds__(256) void k_probe(const double * A, double * out, long long * clocks, int reps)
{
  __shared__ double blk[16 * 18], xd[16 * XDS], rdv[16], sc[64];
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  if (threadIdx.x >= 64) {return;}
  v4d tile;
  for (int r = 0; r < 4; ++r) {tile[r] = A[lr * 16 + lk + 4 * r];}
  bool bad = false;
  long long t0 = clock64();
  for (int it = 0; it < reps; ++it) {
    if (kMode == 0) {
      for (int r = 0; r < 4; ++r) {if (lk + 4 * r <= lr) {blk[lr * 18 + lk + 4 * r] = tile[r];}}
      bad |= potrf_diag(blk, 18, lane, xd, rdv);
    } else {
      bad |= potrf_diag_mfma(tile, blk, 18, lane, xd, rdv, sc);
    }
    tile[0] += blk[0] * 1e-300;           // a dependency from call to call
  }
  long long t1 = clock64();
  if (lane == 0) {clocks[0] = t1 - t0; clocks[1] = bad;}
  for (int i = lane; i < 256; i += 64) {out[i] = blk[(i >> 4) * 18 + (i & 15)]; out[256 + i] = xd[(i >> 4) * XDS + (i & 15)];}
  if (lane < 16) {out[512 + lane] = rdv[lane];}
}

__global__ void k_rsq_error(double * out)
{
  // worst relative error of v_rsq_f64 and of rsqrt_full over a sweep of arguments
  double worst_hw = 0.0, worst_full = 0.0;
  for (int i = 0; i < 20000; ++i) {
    const double d = (1.0 + (threadIdx.x * 20000 + i) * (3.0 / (64 * 20000))) * ((i & 1) ? 1e-6 : 1.0) * ((i & 2) ? 1e5 : 1.0);
    const double exact = 1.0 / sqrt(d);
    worst_hw = fmax(worst_hw, fabs(__builtin_amdgcn_rsq(d) - exact) / exact);
    worst_full = fmax(worst_full, fabs(rsqrt_full(d) - exact) / exact);
  }
  out[threadIdx.x] = worst_hw; out[64 + threadIdx.x] = worst_full;
}

int main()
{
  std::vector<double> A(256), G(16 * 40);
  srand(1);
  for (auto & g : G) {g = (rand() / (double)RAND_MAX) - 0.5;}
  for (int i = 0; i < 16; ++i) {for (int j = 0; j < 16; ++j) {double s = i == j ? 2.0 : 0.0; for (int k = 0; k < 40; ++k) {s += G[i * 40 + k] * G[j * 40 + k];} A[i * 16 + j] = s;}}
  std::vector<double> L(256, 0.0), X(256, 0.0);
  for (int j = 0; j < 16; ++j) {
    double d = A[j * 16 + j]; for (int k = 0; k < j; ++k) {d -= L[j * 16 + k] * L[j * 16 + k];}
    L[j * 16 + j] = std::sqrt(d);
    for (int i = j + 1; i < 16; ++i) {double s = A[i * 16 + j]; for (int k = 0; k < j; ++k) {s -= L[i * 16 + k] * L[j * 16 + k];} L[i * 16 + j] = s / L[j * 16 + j];}
  }
  // X = L^-T: X[i][j] = (L^-1)[j][i]
  std::vector<double> Li(256, 0.0);
  for (int c = 0; c < 16; ++c) {for (int i = c; i < 16; ++i) {double s = i == c ? 1.0 : 0.0; for (int k = c; k < i; ++k) {s -= L[i * 16 + k] * Li[k * 16 + c];} Li[i * 16 + c] = s / L[i * 16 + i];}}
  double * dA; double * dout; long long * dclk;
  hipMalloc(&dA, 256 * 8); hipMalloc(&dout, 1024 * 8); hipMalloc(&dclk, 16);
  hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    for (int threads : {64, 256}) {
      const int reps = 200;
      for (int warm = 0; warm < 2; ++warm) {
        if (mode == 0) {hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(threads), 0, 0, dA, dout, dclk, reps);}
        else {hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(threads), 0, 0, dA, dout, dclk, reps);}
        hipDeviceSynchronize();
      }
      std::vector<double> out(1024); long long clk[2];
      hipMemcpy(out.data(), dout, 1024 * 8, hipMemcpyDeviceToHost); hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
      double eL = 0, eX = 0, eR = 0;
      for (int i = 0; i < 16; ++i) {for (int j = 0; j < 16; ++j) {
        if (j <= i && mode == 0) {eL = std::max(eL, std::fabs(out[i * 16 + j] - L[i * 16 + j]));}
        eX = std::max(eX, std::fabs(out[256 + i * 16 + j] - Li[j * 16 + i]));
      } eR = std::max(eR, std::fabs(out[512 + i] - 1.0 / L[i * 16 + i]));}
      std::printf("%s threads %d: %.0f clocks per block (100 MHz counter ticks x?), bad %lld, err L %.2e X %.2e rd %.2e\n", mode ? "mfma " : "chain", threads,
                  (double)clk[0] / reps, clk[1], eL, eX, eR);
    }
  }
#ifdef STAMPS
  {
    long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(kh::g_dstamp), sizeof(st));
    std::printf("segments per block (gather, chol4, select, panel mfma, update mfma -> next):");
    for (int i = 1; i < 20; ++i) {std::printf(" %lld%s", st[i] - st[i - 1], i % 5 == 0 ? " |" : "");}
    std::printf("\n");
  }
#endif
  hipLaunchKernelGGL(k_rsq_error, dim3(1), dim3(64), 0, 0, dout);
  std::vector<double> e(128); hipMemcpy(e.data(), dout, 128 * 8, hipMemcpyDeviceToHost);
  double hw = 0, full = 0; for (int i = 0; i < 64; ++i) {hw = std::max(hw, e[i]); full = std::max(full, e[64 + i]);}
  std::printf("v_rsq_f64 worst relative error %.3e, rsqrt_full %.3e\n", hw, full);
  return 0;
}
