// Does gfx950 serve a BYTE-misaligned ds_read_b32 (dword at address = 1, 2, 3 mod 4), and at what rate?  K3' (k_score_lds) keeps
// four accumulator sets per wave because its windows start at byte granularity and it reads aligned dwords; if the LDS returned
// the misaligned dword at the aligned rate, one set would do.  Prints, per shift 0..3: whether the data is the bytes at the
// address, and the clocks per wave-level read with 8 waves per CU hammering the port (the pattern of K3': lane -> 16 consecutive
// dwords of a row, four rows per wave instruction, pitch 192 B).
//   hipcc --offload-arch=gfx950 -O2 tools/lds_unaligned_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kPitch = 192, kRows = 200, kBytes = kPitch * kRows;

__global__ __launch_bounds__(512) void probe(int shift, int iters, uint32_t * out, long long * clocks, int b64)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (int i = threadIdx.x; i < kBytes + 64; i += blockDim.x) {lds[i] = (uint8_t)(i * 7 + 3);}
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lx = lane & 15, ly = lane >> 4;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
  uint32_t addr = base + (b64 ? 8 : 4) * lx + (4 * wave + ly) * kPitch + shift;
  uint32_t acc = 0;
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint32_t w[16];
    if (b64) {
      uint64_t v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[r]) : "v"(addr), "n"(r * 16 * kPitch) : "memory");}
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int r = 0; r < 8; ++r) {acc += (uint32_t)v[r] + (uint32_t)(v[r] >> 32);}
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[r]) : "v"(addr), "n"((r & 3) * 16 * kPitch + (r >> 2) * 8) : "memory");}
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
        "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) {acc += w[r];}
    }
    addr += (it & 1) ? -16 : 16;
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) {clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = t1 - t0;}
  // correctness sample: the dword this lane's first address names
  if (blockIdx.x == 0 && iters == 1) {
    const uint32_t a = 4 * lx + (4 * wave + ly) * kPitch + shift;
    uint32_t got;
    const uint32_t a_lds = base + a;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(a_lds) : "memory");
    const uint32_t want = (uint32_t)lds[a] | ((uint32_t)lds[a + 1] << 8) | ((uint32_t)lds[a + 2] << 16) | ((uint32_t)lds[a + 3] << 24);
    out[threadIdx.x] = got == want ? 1u : 0u;
  }
}

int main()
{
  uint32_t * d_out; long long * d_clk;
  const int blocks = 512;
  hipMalloc(&d_out, blocks * 512 * 4); hipMalloc(&d_clk, blocks * 16);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes + 64);
  std::vector<uint32_t> h(512); std::vector<long long> c(2 * blocks);
  for (int b64 = 0; b64 < 2; ++b64) {
    for (int shift = 0; shift < (b64 ? 8 : 4); ++shift) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(512), kBytes + 64, 0, shift, 1, d_out, d_clk, 0);
      hipMemcpy(h.data(), d_out, 512 * 4, hipMemcpyDeviceToHost);
      int ok = 0; for (uint32_t v : h) {ok += v == 1u;}
      const int iters = 4000;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), kBytes + 64, 0, shift, iters, d_out, d_clk, b64);
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), kBytes + 64, 0, shift, iters, d_out, d_clk, b64);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(c.data(), d_clk, blocks * 16, hipMemcpyDeviceToHost);
      double cyc = 0; for (int i = 0; i < blocks; ++i) {cyc += c[2 * i];}
      cyc /= blocks;
      const double reads = (double)iters * (b64 ? 8 : 16);
      // bytes all workgroups read / time
      const double tbs = (double)blocks * 8 * reads * 64 * (b64 ? 8 : 4) / (ms * 1e-3) / 1e12;
      printf("%s shift %d: b32 data %s (%d/512 lanes); %.1f shader clocks per wave read (8 waves per workgroup), launch %.3f ms = %.1f TB/s\n",
        b64 ? "ds_read_b64" : "ds_read_b32", shift, ok == 512 ? "correct" : "WRONG", ok, cyc / reads, ms, tbs);
    }
  }
  return 0;
}
