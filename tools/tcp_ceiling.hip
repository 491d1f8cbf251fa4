// Ceiling of the vector L1 (TCP) for the load pattern of k_score: a wave-level dword load whose 64 lanes read FOUR 64-byte
// row segments (16 lanes x 4 B each, one per lane group) of a buffer that stays resident in the CU's L1.  T = number of
// 128-byte lines the four segments touch: 4 when every segment lies inside one line, up to 8 when every segment straddles
// a line boundary.  Round 2's model for k_score was  cycles per wave-level load = T (tag look-ups) + 4 (64-byte data
// beats); this measures it, for windows that hit the L1 and for windows that miss it and hit the L2:  hipcc --offload-arch=gfx950 -O3 tools/tcp_ceiling.hip -o /tmp/tcp_ceiling && /tmp/tcp_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

// rows of `pitch` dwords; lane group g = lane >> 4 reads row (4 * r + g), 16 consecutive dwords from column `col`.
// straddle: bit g set -> group g's segment starts 24 dwords into a 32-dword (128-byte) line, i.e. covers two lines.
template <int UNROLL>
__global__ __launch_bounds__(512) void k_probe(const unsigned * __restrict__ buf, int straddle, int iters, int halves, unsigned * out)
{
  constexpr int kPitch = 64;                  // dwords per row (256 B): a 64-byte segment at column 0 sits inside one 128-byte line
  const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
  const int col = ((straddle >> g) & 1) ? 24 : 0;
  unsigned acc[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {acc[u] = 0;}
  // every workgroup walks its own window: two halves of 32 rows x 256 B = 8 KB each, taken in turn so that the addresses
  // change with the iteration (nothing to hoist) while the 16 KB stay resident in the CU's 32 KB L1
  const unsigned * base = buf + (size_t)blockIdx.x * ((size_t)halves * 32 * kPitch) + (((threadIdx.x >> 6) & 7) * 4 + g) * kPitch + col + l;
  for (int it = 0; it < iters; ++it) {
    const unsigned * p = base + ((it & (halves - 1)) << 11);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {acc[u] += p[((u * 4 * kPitch) & 2047)];}          // row groups 0..7 of the half, wrapped
  }
  unsigned s = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {s += acc[u];}
  if (s == 0x12345678u) {out[0] = s;}          // never true: keeps the loads alive
}

// the same for dwordx2 loads: lane = (row lane >> 3 of 8, 8-byte chunk lane & 7): EIGHT 64-byte row segments per wave-level load
// (512 bytes); `shift` = 4 moves every segment to a dword-but-not-qword-aligned start (the alignment classes of k_score)
template <int UNROLL>
__global__ __launch_bounds__(512) void k_probe2(const unsigned * __restrict__ buf, int shift_words, int iters, int halves, unsigned * out)
{
  constexpr int kPitch = 64;
  const int lane = threadIdx.x & 63, g = lane >> 3, l = lane & 7;
  unsigned long long acc[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {acc[u] = 0;}
  // a wave covers 8 rows per load; 8 waves x 8 loads: the 32 rows of a half are visited twice per iteration
  const unsigned * base = buf + (size_t)blockIdx.x * ((size_t)halves * 32 * kPitch) + (((threadIdx.x >> 6) & 3) * 8 + g) * kPitch + shift_words + 2 * l;
  for (int it = 0; it < iters; ++it) {
    const unsigned * p = base + ((it & (halves - 1)) << 11);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      unsigned lo, hi;
      const unsigned * q = p + ((u & 1) * 16);            // two column positions inside the 256-byte row
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(*(unsigned long long *)&acc[u]) : "v"(q) : "memory");
      (void)lo; (void)hi;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long s = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {s += acc[u];}
  if (s == 0x12345678ull) {out[0] = (unsigned)s;}
}

// mixed hits and misses: of the four 64-byte row segments of a wave-level dword load, `miss_groups` come from a window that is
// evicted before its next use (L1 miss, L2 hit: 64 KB per workgroup, walked round robin) and the rest from a resident one
template <int UNROLL>
__global__ __launch_bounds__(512) void k_probe_mix(const unsigned * __restrict__ buf, int miss_groups, int iters, unsigned * out)
{
  constexpr int kPitch = 64;
  const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15;
  const int wave = (threadIdx.x >> 6) & 7;
  unsigned acc[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {acc[u] = 0;}
  // per workgroup: 8 halves of 8 KB for the streaming part, then 8 KB resident (4 x 8 rows x 256 B)
  const unsigned * wg = buf + (size_t)blockIdx.x * (9 * 32 * kPitch);
  const unsigned * hot = wg + 8 * 32 * kPitch + (wave * 4 + g) * kPitch + l;
  const unsigned * cold = wg + (wave * 4 + g) * kPitch + l;
  const bool miss = g < miss_groups;
  for (int it = 0; it < iters; ++it) {
    const unsigned * p = miss ? cold + ((it & 7) << 11) : hot;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      // the resident rows are re-read with a column shift inside the line (no new line), the streaming ones walk the half
      const unsigned * q = miss ? p + ((u * 4 * kPitch) & 2047) : p + 16 * (u & 1);
      acc[u] += *q;
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {s += acc[u];}
  if (s == 0x12345678u) {out[0] = s;}
}

int main()
{
  const int n_cu = 256, threads = 512;                          // one workgroup of 8 waves per CU
  const int blocks = n_cu;
  unsigned * d = nullptr, * o = nullptr;
  const size_t words = (size_t)blocks * 16 * 32 * 64 + 4096;
  hipMalloc(&d, words * 4); hipMemset(d, 1, words * 4); hipMalloc(&o, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, unroll = 8;
  // window per workgroup = halves x 8 KB, walked round robin: 16 KB stays in the CU's 32 KB L1 (every load hits); 64 KB
  // does not (every line is evicted before its next use: every load misses the L1 and hits the L2, 16 MB over 8 XCDs)
  std::printf("# window_KB  lines_per_load  ms  wave_loads  cycles_per_wave_load_per_CU(2.4GHz)  TB/s_to_registers\n");
  const int masks[5] = {0x0, 0x1, 0x3, 0x7, 0xF};
  const int halves_of[3] = {2, 8, 16};
  for (int h = 0; h < 3; ++h) {
    for (int k = 0; k < 5; ++k) {
      const int tags = 4 + k, halves = halves_of[h];
      hipLaunchKernelGGL(k_probe<8>, dim3(blocks), dim3(threads), 0, 0, d, masks[k], 200, halves, o);   // warm the caches / clocks
      hipDeviceSynchronize();
      float best = 1e30f;
      const int n_it = h == 0 ? iters : iters / 4;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe<8>, dim3(blocks), dim3(threads), 0, 0, d, masks[k], n_it, halves, o);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {best = ms;}
      }
      const double wave_loads = (double)blocks * (threads / 64) * n_it * unroll;
      const double cyc = best * 1e-3 * 2.4e9 * n_cu / wave_loads;
      std::printf("%d  %d  %.3f  %.3e  %.2f  %.1f\n", halves * 8, tags, best, wave_loads, cyc, wave_loads * 256.0 / (best * 1e-3) / 1e12);
    }
  }
  std::printf("# mixed: of the 4 row segments (lines) of a dword load, N miss the L1 and hit the L2\n# missing_lines  ms  wave_loads  cycles_per_wave_load_per_CU  TB/s_to_registers\n");
  for (int m = 0; m <= 4; ++m) {
    hipLaunchKernelGGL(k_probe_mix<8>, dim3(blocks), dim3(threads), 0, 0, d, m, 200, o);
    hipDeviceSynchronize();
    float best = 1e30f;
    const int n_it = iters / 4;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_probe_mix<8>, dim3(blocks), dim3(threads), 0, 0, d, m, n_it, o);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) {best = ms;}
    }
    const double wave_loads = (double)blocks * (threads / 64) * n_it * unroll;
    const double cyc = best * 1e-3 * 2.4e9 * n_cu / wave_loads;
    std::printf("%d  %.3f  %.3e  %.2f  %.1f\n", m, best, wave_loads, cyc, wave_loads * 256.0 / (best * 1e-3) / 1e12);
  }
  std::printf("# dwordx2 loads, 8 row segments of 64 B per wave-level load (512 B)\n# window_KB  start_mod_8  ms  wave_loads  cycles_per_wave_load_per_CU  TB/s_to_registers\n");
  for (int h = 0; h < 2; ++h) {
    for (int sh = 0; sh < 2; ++sh) {
      const int halves = halves_of[h];
      hipLaunchKernelGGL(k_probe2<8>, dim3(blocks), dim3(threads), 0, 0, d, sh, 200, halves, o);
      hipDeviceSynchronize();
      float best = 1e30f;
      const int n_it = h == 0 ? iters : iters / 4;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe2<8>, dim3(blocks), dim3(threads), 0, 0, d, sh, n_it, halves, o);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {best = ms;}
      }
      const double wave_loads = (double)blocks * (threads / 64) * n_it * unroll;
      const double cyc = best * 1e-3 * 2.4e9 * n_cu / wave_loads;
      std::printf("%d  %d  %.3f  %.3e  %.2f  %.1f\n", halves * 8, 4 * sh, best, wave_loads, cyc, wave_loads * 512.0 / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
