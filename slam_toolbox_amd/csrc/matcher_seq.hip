// gfx950 kernels of the fused path of ONE MatchScan (see matcher_seq.hpp for the plan).  Same arithmetic as the batch kernels
// (matcher_kernels.hip), bit for bit; what differs is the layout of the work for a single job on a nearly empty chip: no
// hash table, state of the order-dependent rule in LDS, one workgroup where the batch path takes five launches, table and
// scoring in one launch with the beams cut into slices, the fine pass finished on the device.
// Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "kh_internal.hpp"
#include "matcher_device.hpp"
#include "matcher_seq.hpp"

namespace kh
{

// ---------------------------------------------------------------------------------------------
// kseq_prep.  One workgroup per (job, base scan): FindValidPoints of the scan (Mapper.cpp:1113-1164; find_valid_scan), then --
// the scan's points and flags still in LDS -- WorldToGrid + the ROI test of AddScan (Mapper.cpp:1083-1088) for every kept
// reading and the FIRST point of every cell: atomicMin of the job point index into first[cell].  A second point of a cell
// stamps the same footprint (n_foot == 0), or is skipped by the "cell already 100" rule whatever happened to the first
// (n_foot > 0): only firsts are stamp candidates.  Workgroups behind them: Grid::Clear (Karto.h:4612-4615) = the tiles the
// slot's previous rasterisation wrote, the counters.  One job: everything arrives as kernel arguments; a batch: job
// descriptors and (job, scan) items in device memory, as for k_find_valid_par.
__device__ __forceinline__ void prep_scan(const RasterJob & job, const double2 * pts, int n, int p0, int max_n, double2 * s_fv, long long * dbg)
{
  uint8_t * flags = nullptr;
  find_valid_scan<false>(pts, n, job.active + p0, job.view_x, job.view_y, max_n, s_fv, flags, dbg);
  const double2 * P = s_fv;
  int32_t * const first = job.first;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int32_t gx = 0, gy = 0;
    bool on = flags[i] != 0;
    if (on) {on = roi_cell(job, P[i], gx, gy);}
    int2 cell = make_int2(-1, -1);
    if (on) {
      cell = make_int2(gx + job.roi_x, gy + job.roi_y);              // CorrelationGrid::GridIndex, Mapper.h:1122-1128
      atomicMin(&first[(size_t)gy * job.roi_w + gx], p0 + i);
    }
    *reinterpret_cast<int2 *>(job.cell_xy + 2 * (size_t)(p0 + i)) = cell;
  }
  if (dbg && threadIdx.x == 0) {dbg[6] = (long long)wall_clock64();}
}
// Grid::Clear: one WAVE per tile the previous rasterisation wrote (lane = tile row, 64 bytes each) -- a workgroup walking its
// tiles one after the other waits a memory latency per tile for the tile's index
__device__ __forceinline__ void prep_clear(const RasterJob & job, int c, int clear_blocks)
{
  const int n_prev = job.prev_work[0];
  const int waves = clear_blocks * (int)(blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  for (int w = c * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6); w < n_prev; w += waves) {
    const int t = job.prev_work[4 + w];
    const int ty = t / job.tiles_w, tx = t - ty * job.tiles_w;
    const int y = ty * kRasterTile + lane, x = tx * kRasterTile;
    if (y < job.height) {
      uint8_t * at = job.grid + (size_t)y * job.ws + x;           // ws is a multiple of 8, the tile starts on a multiple of 64
      const int nb = min(kRasterTile, job.ws - x);
      for (int i = 0; i < nb; i += 8) {*reinterpret_cast<uint2 *>(at + i) = make_uint2(0u, 0u);}
    }
  }
  if (c == 0) {
    if (threadIdx.x < 4) {job.n_work[threadIdx.x] = 0;}
    if (threadIdx.x < kSeqCtlWords) {job.seq_ctl[threadIdx.x] = 0;}
  }
}

__global__ __launch_bounds__(1024) void kseq_prep(const SeqPrepArgs args)
{
  extern __shared__ double2 s_fv[];
  const RasterJob & job = args.job;
  const int b = blockIdx.x;
  if (b < args.n_scans) {
    const int p0 = args.prefix[b];
    prep_scan(job, reinterpret_cast<const double2 *>(args.scans[b]), args.prefix[b + 1] - p0, p0, args.max_n, s_fv, b == 0 ? args.dbg : nullptr);
    return;
  }
  const int c = b - args.n_scans;
  prep_clear(job, c, args.clear_blocks);
  if (c == 0) {
    // the job for the launches that follow
    const uint32_t * src = reinterpret_cast<const uint32_t *>(&args.job);
    uint32_t * dst = reinterpret_cast<uint32_t *>(args.d_job);
    for (int i = threadIdx.x; i < (int)(sizeof(RasterJob) / 4); i += blockDim.x) {dst[i] = src[i];}
  }
}

__global__ __launch_bounds__(1024) void kseq_prep_batch(const RasterJob * jobs, const ValidItem * items, int n_items, int max_n, int clear_blocks)
{
  extern __shared__ double2 s_fv[];
  const int b = blockIdx.x;
  if (b < n_items) {
    const RasterJob & job = jobs[items[b].job];
    const int k = items[b].scan;
    const int p0 = job.scan_prefix[k];
    prep_scan(job, reinterpret_cast<const double2 *>(job.scan_ptr[k]), job.scan_prefix[k + 1] - p0, p0, max_n, s_fv, nullptr);
    return;
  }
  const int c = b - n_items;
  prep_clear(jobs[c / clear_blocks], c % clear_blocks, clear_blocks);
}

static size_t prep_lds_bytes(int max_n)
{
  const size_t stride_i = (size_t)max_n + 64;
  return (size_t)max_n * sizeof(double2) + 3 * stride_i * sizeof(int32_t) + 2 * stride_i;
}

void launch_seq_prep(const SeqPrepArgs & args, void * stream)
{
  static std::atomic<unsigned long long> done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(kseq_prep), 64 * 1024, done);
  hipLaunchKernelGGL(kseq_prep, dim3(args.n_scans + args.clear_blocks), dim3(1024), prep_lds_bytes(args.max_n), (hipStream_t)stream, args);
}

// a batch: 512 threads per (job, scan) -- thousands of scans keep the chip busy without the extra waves
void launch_seq_prep_batch(const RasterJob * d_jobs, int32_t n_jobs, const ValidItem * d_items, int32_t n_items, int32_t max_n, void * stream)
{
  if (n_jobs <= 0) {return;}
  static std::atomic<unsigned long long> done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(kseq_prep_batch), 64 * 1024, done);
  const int clear_blocks = 16;                // per job: 64 waves a pass
  // (256 threads per item: 385 us per 6000 items, 512: 352, 1024: 428)
  hipLaunchKernelGGL(kseq_prep_batch, dim3(n_items + n_jobs * clear_blocks), dim3(512), prep_lds_bytes(max_n), (hipStream_t)stream, d_jobs, d_items,
    (int)n_items, (int)max_n, clear_blocks);
}

// ---------------------------------------------------------------------------------------------
// kseq_links.  Thread per job point: the firsts become stamp candidates (point, cell); with the order-dependent rule
// (Mapper.cpp:1093-1096, n_foot > 0) each takes along the first points of the cells in its 100-footprint that come EARLIER
// (a later one can never block it) -- four independent loads where the batch path probes a hash table.
__global__ __launch_bounds__(256) void kseq_links(const RasterJob * jobs)
{
  // (a COPY: its fields are fetched in one batch of scalar loads here; through a reference every field was fetched at its first use,
  // each behind a wait of its own -- a dozen dependent round trips in kernels that live for five to twenty microseconds)
  const RasterJob job = jobs[blockIdx.y];
  const int32_t * __restrict__ first = job.first;
  int32_t * cand = job.cand;
  int32_t * ctl = job.seq_ctl;
  const int p = blockIdx.x * 256 + threadIdx.x;
  bool is_first = false;
  int cx = -1, cy = -1;
  int nb[kMaxFootprint] = {-1, -1, -1, -1};
  if (p < job.n_points) {
    const int2 c = *reinterpret_cast<const int2 *>(job.cell_xy + 2 * (size_t)p);
    cx = c.x; cy = c.y;
    if (cx >= 0) {
      const int gx = cx - job.roi_x, gy = cy - job.roi_y;
      int32_t v[kMaxFootprint] = {kFirstNone, kFirstNone, kFirstNone, kFirstNone};
      const int32_t mine = first[(size_t)gy * job.roi_w + gx];
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {
        if (f < job.n_foot) {
          const int nx = gx + job.foot_dx[f], ny = gy + job.foot_dy[f];
          if (nx >= 0 && nx < job.roi_w && ny >= 0 && ny < job.roi_h) {v[f] = first[(size_t)ny * job.roi_w + nx];}
        }
      }
      is_first = mine == p;
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {nb[f] = v[f] < p ? v[f] : -1;}
    }
  }
  // candidate list: one atomic per wave
  const unsigned long long mask = __ballot(is_first);
  if (mask == 0) {return;}
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(mask);
  int base = 0;
  if (lane == leader) {base = atomicAdd(&ctl[0], __builtin_popcountll(mask));}
  base = __shfl(base, leader);
  if (is_first) {
    const int i = base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
    int4 * rec = reinterpret_cast<int4 *>(cand + (size_t)kSeqCandWords * i);
    rec[0] = make_int4(p, cx, cy, 0);
    rec[1] = make_int4(nb[0], nb[1], nb[2], nb[3]);
  }
}

void launch_seq_links(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, void * stream)
{
  if (n_jobs <= 0 || max_points <= 0) {return;}
  hipLaunchKernelGGL(kseq_links, dim3((max_points + 255) / 256, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs);
}

// ---------------------------------------------------------------------------------------------
// exclusive prefix sum over the 1024 threads of a block (two 32-bit sums packed in one 64-bit word); total to every thread
__device__ __forceinline__ unsigned long long block_exscan_1024(unsigned long long v, unsigned long long * s_w, unsigned long long & total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long o = __shfl_up(inc, d);
    if (lane >= d) {inc += o;}
  }
  if (lane == 63) {s_w[wave] = inc;}
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const unsigned long long x = s_w[w];
    if (w < wave) {base += x;}
    tot += x;
  }
  total = tot;
  return base + inc - v;
}

// kseq_bin.  ONE workgroup of 1024 threads does what the batch path spreads over k_active_set, k_raster_bin, k_raster_scan,
// k_raster_fill and k_repitch_keep -- for one job those are five launches of dependent round trips to L2 on a handful of
// compute units; here the state lives in LDS and the candidates (the first 8192: tid + 1024 k) in registers from the first
// phase to the last:
//  (the first-point table is handed back clean by the launch that follows: stage_copy)
//  1. the order-dependent rule: a candidate is stamped iff no EARLIER STAMPED candidate has its cell in its 100-footprint -- the
//     greedy independent set in point order (see k_active_set).  State byte per job point in LDS (0 undecided, 1 stamped,
//     2 skipped); a candidate decides as soon as its earlier neighbours have.  NO barrier between the rounds: decisions never
//     change, a stale read only delays, and every wave of the workgroup is resident -- the owner of the earliest undecided
//     candidate always runs -- so each thread simply polls until its own candidates are decided (a barrier per round made the
//     dependency chains along walls, a hundred cells long, cost a hundred barriers: 20 of the kernel's 47 us).
//  2. every stamped candidate marks the occupancy blocks its footprint overlaps and takes its rank in the <= 2 x 2 tiles it
//     overlaps from LDS counters; scan of the counters -> list starts, the list of non-empty tiles (which also becomes the
//     "previous" list Grid::Clear of the next match zeroes); the lists are written from the ranks.
constexpr int kBinRegs = 8;
__global__ __launch_bounds__(1024) void kseq_bin(const RasterJob * jobs, int bm_global, long long * dbg)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
  __shared__ unsigned long long s_w[16];
  // everything the kernel reads from the job, once: behind a store the compiler has to assume the job block itself changed and
  // re-reads every field through the scalar cache -- a wait per field and loop trip
  struct {int n_points, n_foot, tiles_w, tiles_h, bm_w, bm_h, bshift, roi_x, roi_y, roi_w, kernel_size;
          int32_t * rank, * list, * tile_start, * tile_count, * work, * prev_work, * n_work; uint32_t * blockmap;} job;
  int32_t * cand;
  const int32_t * ctl;
  int4 * work2;
  int keep_prev;
  {
    const RasterJob & j = jobs[blockIdx.x];
    cand = j.cand; ctl = j.seq_ctl; work2 = reinterpret_cast<int4 *>(j.work2);
    keep_prev = j.grid2 == nullptr ? 1 : 0;          // a slot with copies of its grid: k_repitch* read the previous list first, then keep this one
    job.n_points = j.n_points; job.n_foot = j.n_foot; job.tiles_w = j.tiles_w; job.tiles_h = j.tiles_h; job.bm_w = j.bm_w; job.bm_h = j.bm_h; job.bshift = j.bshift;
    job.roi_x = j.roi_x; job.roi_y = j.roi_y; job.roi_w = j.roi_w; job.kernel_size = j.kernel_size;
    job.rank = j.rank; job.list = j.list; job.tile_start = j.tile_start; job.tile_count = j.tile_count; job.work = j.work;
    job.prev_work = j.prev_work; job.n_work = j.n_work; job.blockmap = j.blockmap;
  }
  const int tid = threadIdx.x;
  const int n_cand = ctl[0];
  int stamp = 0;
  auto phase = [&]() {if (dbg && tid == 0) {dbg[stamp] = (long long)wall_clock64();} ++stamp;};
  phase();
  const int tiles = job.tiles_w * job.tiles_h, bm_words = job.bm_w * job.bm_h;
  const int state_bytes = job.n_foot > 0 ? ((job.n_points + 15) & ~15) : 0;
  // (an LDS pointer by type: through a generic one every volatile read of a state byte is a FLAT load with a full wait behind it)
  typedef __attribute__((address_space(3))) volatile uint8_t lds_state;
  lds_state * const state = (lds_state *)(__attribute__((address_space(3))) uint8_t *)s_dyn;
  int32_t * s_cnt = reinterpret_cast<int32_t *>(s_dyn + state_bytes);
  // the occupancy block map is built in LDS and written out once -- or, when it does not fit beside the rest (8 x 8-cell blocks
  // of a large grid: 134 KB for the config-2 geometry), marked in place
  uint32_t * s_bm = bm_global ? job.blockmap : reinterpret_cast<uint32_t *>(s_cnt + tiles);
  for (int i = tid; i < tiles; i += 1024) {s_cnt[i] = 0;}
  for (int i = tid; i < bm_words; i += 1024) {s_bm[i] = 0u;}
  const int4 * rec = reinterpret_cast<const int4 *>(cand);
  const int extra0 = 1024 * kBinRegs;                        // candidates from here on (rare) go through memory
  int pr[kBinRegs], cxy[kBinRegs];
  int4 nbr[kBinRegs];
  // a thread's candidates are NEIGHBOURS in the list: kseq_links appends wave by wave, each wave's firsts in point order, and the
  // dependency chains of the rule run along walls in point order -- so a chain link is mostly decided by the thread that decided
  // the link before it, in the same trip of its loop, instead of waiting for another wave's next trip
  const int per = min(kBinRegs, (n_cand + 1023) / 1024);
  {
    // (all of a thread's records requested together -- unconditional loads at a clamped index: as `if (mine) {load}` every record sat
    // in a branch of its own with a wait behind it, seven dependent round trips in front of the first verdict)
    typedef int rec4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const rec4 grec;
    grec * const grecs = (grec *)rec;
    rec4 ra[kBinRegs], rb[kBinRegs];
    const int last = max(n_cand - 1, 0);
#pragma unroll
    for (int k = 0; k < kBinRegs; ++k) {
      const int ic = min(tid * per + k, last);
      ra[k] = grecs[2 * (size_t)ic]; rb[k] = grecs[2 * (size_t)ic + 1];
    }
#pragma unroll
    for (int k = 0; k < kBinRegs; ++k) {
      const int i = tid * per + k;
      const bool have = k < per && i < n_cand;
      pr[k] = have ? ra[k].x : -1; cxy[k] = have ? (ra[k].y | (ra[k].z << 16)) : 0;
      nbr[k] = (have && job.n_foot > 0) ? make_int4(rb[k].x, rb[k].y, rb[k].z, rb[k].w) : make_int4(-1, -1, -1, -1);
    }
  }
  if (job.n_foot > 0) {
#pragma unroll
    for (int k = 0; k < kBinRegs; ++k) {if (pr[k] >= 0) {state[pr[k]] = 0;}}
    for (int i = tid + extra0; i < n_cand; i += 1024) {state[rec[2 * (size_t)i].x] = 0;}
    __syncthreads();
    // a candidate's verdict from the states of its earlier neighbours (0 = still undecided)
    auto verdict = [&](const uint8_t (&st)[kMaxFootprint]) -> uint8_t {
      bool blocked = false, waiting = false;
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {blocked = blocked || st[f] == 1; waiting = waiting || st[f] == 0;}
      return blocked ? (uint8_t)2 : (waiting ? (uint8_t)0 : (uint8_t)1);
    };
    auto decide = [&](int p, const int4 nb) -> bool {             // true = still undecided (candidates beyond the registers)
      const int n4[4] = {nb.x, nb.y, nb.z, nb.w};
      uint8_t st[kMaxFootprint];
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {st[f] = n4[f] >= 0 ? state[n4[f]] : (uint8_t)2;}
      const uint8_t v = verdict(st);
      if (v) {state[p] = v;}
      return v == 0;
    };
    uint32_t undecided = 0;
    uint8_t mine[kBinRegs];                                      // this thread's own verdicts: a chain link whose predecessor is the
#pragma unroll                                                   // thread's previous candidate reads it here, not through LDS
    for (int k = 0; k < kBinRegs; ++k) {mine[k] = 0; if (pr[k] >= 0) {undecided |= 1u << k;}}
    bool extra_left = tid + extra0 < n_cand;
    for (int spin = 0; (undecided != 0 || extra_left) && spin < (1 << 24); ++spin) {
#pragma unroll
      for (int k = 0; k < kBinRegs; ++k) {
        if ((undecided >> k) & 1u) {
          const int n4[4] = {nbr[k].x, nbr[k].y, nbr[k].z, nbr[k].w};
          uint8_t st[kMaxFootprint];
#pragma unroll
          for (int f = 0; f < kMaxFootprint; ++f) {
            st[f] = 2;
            if (n4[f] >= 0) {
              if (k > 0 && n4[f] == pr[k > 0 ? k - 1 : 0]) {st[f] = mine[k > 0 ? k - 1 : 0];}
              else if (k > 1 && n4[f] == pr[k > 1 ? k - 2 : 0]) {st[f] = mine[k > 1 ? k - 2 : 0];}
              else {st[f] = state[n4[f]];}
            }
          }
          const uint8_t v = verdict(st);
          if (v) {mine[k] = v; state[pr[k]] = v; undecided &= ~(1u << k);}
        }
      }
      if (extra_left) {
        extra_left = false;
        for (int i = tid + extra0; i < n_cand; i += 1024) {
          const int p = rec[2 * (size_t)i].x;
          if (state[p] == 0 && decide(p, rec[2 * (size_t)i + 1])) {extra_left = true;}
        }
      }
    }
  }
  __syncthreads();
  phase();
  const int hk = job.kernel_size / 2;
  auto tile_of = [&](int cx, int cy, int q) -> int {
    const int tx0 = (cx - hk) / kRasterTile, tx1 = (cx + hk) / kRasterTile;
    const int ty0 = (cy - hk) / kRasterTile, ty1 = (cy + hk) / kRasterTile;
    const int tx = tx0 + (q & 1), ty = ty0 + (q >> 1);
    return (tx <= tx1 && ty <= ty1) ? ty * job.tiles_w + tx : -1;
  };
  // one candidate: its cell goes back to "none" in the first-point table; if it is stamped: occupancy blocks, ranks in its tiles
  auto count_one = [&](int p, int cx, int cy, int (&rk)[4]) -> bool {
    rk[0] = rk[1] = rk[2] = rk[3] = -1;
    if (job.n_foot > 0 && state[p] != 1) {return false;}
    const int fx0 = (cx - hk) >> job.bshift, fx1 = (cx + hk) >> job.bshift;
    const int fy0 = (cy - hk) >> job.bshift, fy1 = (cy + hk) >> job.bshift;
    // the footprint's blocks of a block row are neighbouring bits: one mask per row and word (two words where the run of bits
    // crosses a word end) instead of a probe per block -- a 19-cell footprint on 8 x 8 blocks is 3-4 blocks a side
    const int w0 = fx0 >> 5, w1 = fx1 >> 5;
    const uint32_t m0 = w0 == w1 ? (uint32_t)(((1ull << (fx1 - fx0 + 1)) - 1ull) << (fx0 & 31)) : ~0u << (fx0 & 31);
    const uint32_t m1 = w0 == w1 ? 0u : (uint32_t)((1ull << ((fx1 & 31) + 1)) - 1ull);
    for (int by = fy0; by <= fy1; ++by) {
      uint32_t * word = &s_bm[by * job.bm_w + w0];
      // in LDS: test first (the bits are set already for all but the first stamps of a block); in global memory the test would be
      // a memory round trip per row -- the atomics leave the wave without waiting for anything
      if (bm_global) {atomicOr(word, m0);}
      else if ((*reinterpret_cast<volatile uint32_t *>(word) & m0) != m0) {atomicOr(word, m0);}
      if (m1) {
        if (bm_global) {atomicOr(word + 1, m1);}
        else if ((*reinterpret_cast<volatile uint32_t *>(word + 1) & m1) != m1) {atomicOr(word + 1, m1);}
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = tile_of(cx, cy, q);
      if (t >= 0) {rk[q] = atomicAdd(&s_cnt[t], 1);}
    }
    return true;
  };
  int rks[kBinRegs][4];
  uint32_t stamped = 0;
#pragma unroll
  for (int k = 0; k < kBinRegs; ++k) {
    rks[k][0] = rks[k][1] = rks[k][2] = rks[k][3] = -1;
    if (pr[k] >= 0 && count_one(pr[k], cxy[k] & 0xffff, cxy[k] >> 16, rks[k])) {stamped |= 1u << k;}
  }
  for (int i = tid + extra0; i < n_cand; i += 1024) {
    const int4 a = rec[2 * (size_t)i];
    int rk[4];
    if (count_one(a.x, a.y, a.z, rk)) {
      cand[(size_t)kSeqCandWords * i + 3] = 1;
      *reinterpret_cast<int4 *>(job.rank + 4 * (size_t)a.x) = make_int4(rk[0], rk[1], rk[2], rk[3]);
    }
  }
  __syncthreads();
  phase();
  // list starts and the list of non-empty tiles: each thread a run of consecutive tiles
  {
    const int per = (tiles + 1023) / 1024;
    const int lo = min(tiles, tid * per), hi = min(tiles, lo + per);
    unsigned int sum = 0, nonempty = 0;
    for (int t = lo; t < hi; ++t) {const int c = s_cnt[t]; sum += (unsigned int)c; nonempty += c > 0 ? 1u : 0u;}
    unsigned long long total = 0;
    const unsigned long long before = block_exscan_1024(((unsigned long long)nonempty << 32) | sum, s_w, total);
    int run = (int)(before & 0xffffffffull), wpos = (int)(before >> 32);
    for (int t = lo; t < hi; ++t) {
      const int c = s_cnt[t];
      if (c > 0) {
        job.tile_start[t] = run; job.tile_count[t] = c;
        job.work[wpos] = t;
        if (work2) {work2[wpos] = make_int4(t, run, c, 0);}         // kseq_tile: tile, list start, count in ONE load
        if (keep_prev) {job.prev_work[4 + wpos] = t;}
        ++wpos;
      }
      s_cnt[t] = run;
      run += c;
    }
    if (tid == 0) {
      job.n_work[0] = (int)(total >> 32);
      if (keep_prev) {job.prev_work[0] = (int)(total >> 32);}
    }
  }
  __syncthreads();
  phase();
  // the lists, from the ranks (kseq_tile reads the cell from the entry itself)
  auto fill_one = [&](int p, int cx, int cy, const int (&rk)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = tile_of(cx, cy, q);
      if (t >= 0) {job.list[s_cnt[t] + rk[q]] = work2 ? (cx | (cy << 16)) : p;}
    }
  };
#pragma unroll
  for (int k = 0; k < kBinRegs; ++k) {
    if ((stamped >> k) & 1u) {fill_one(pr[k], cxy[k] & 0xffff, cxy[k] >> 16, rks[k]);}
  }
  for (int i = tid + extra0; i < n_cand; i += 1024) {
    if (cand[(size_t)kSeqCandWords * i + 3] == 0) {continue;}
    const int4 a = rec[2 * (size_t)i];
    const int4 r4 = *reinterpret_cast<const int4 *>(job.rank + 4 * (size_t)a.x);
    const int rk[4] = {r4.x, r4.y, r4.z, r4.w};
    fill_one(a.x, a.y, a.z, rk);
  }
  if (!bm_global) {for (int i = tid; i < bm_words; i += 1024) {job.blockmap[i] = s_bm[i];}}
  phase();
  if (dbg && tid == 0) {dbg[stamp] = n_cand;}
}

size_t seq_bin_lds_bytes(int32_t n_points, int32_t n_foot, int32_t tiles, int32_t bm_words)
{
  const size_t state_bytes = n_foot > 0 ? (((size_t)n_points + 15) & ~(size_t)15) : 0;
  return state_bytes + 4 * (size_t)tiles + 4 * (size_t)std::max(bm_words, 0) + 16;        // (bm_words = 0: the block map stays in global memory)
}

int launch_seq_bin(const RasterJob * d_jobs, int32_t n_jobs, size_t lds_bytes, int32_t bm_global, long long * dbg, void * stream)
{
  if (n_jobs <= 0) {return 0;}
  static std::atomic<unsigned long long> done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(kseq_bin), 158 * 1024, done);
  hipLaunchKernelGGL(kseq_bin, dim3(n_jobs), dim3(1024), lds_bytes, (hipStream_t)stream, d_jobs, (int)bm_global, dbg);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The host's tables (pinned, host-coherent) into device memory -- one read of each byte over the link instead of one per
// workgroup that uses them -- and the sums volume / result block of the coarse pass zeroed.  Runs as the last workgroups of the
// stamping launch (the host makes the tables while kseq_prep .. kseq_bin run, and nothing reads them before kseq_score), or
// as a launch of its own in front of the batch path's stamping kernel.
// ... and the first-point table handed back clean: every cell a rasterisation touched has exactly one candidate.
struct SeqStage {const uint4 * src; uint4 * dst; int units; int32_t * sums; int n_sums; unsigned long long * out; int out_words;};
__device__ __forceinline__ void stage_copy(const SeqStage & g, const RasterJob & job, int tid, int nth)
{
  const int n_cand = job.seq_ctl[0];
  const int roi_x = job.roi_x, roi_y = job.roi_y, roi_w = job.roi_w;
  int32_t * const first = job.first;
  const int4 * const rec = reinterpret_cast<const int4 *>(job.cand);
  for (int i = tid; i < g.units; i += nth) {g.dst[i] = g.src[i];}
  for (int i = tid; i < g.n_sums; i += nth) {g.sums[i] = 0;}
  for (int i = tid; i < g.out_words; i += nth) {g.out[i] = 0ull;}
  for (int i = tid; i < n_cand; i += nth) {
    const int4 a = rec[2 * (size_t)i];
    first[(size_t)(a.z - roi_y) * roi_w + (a.y - roi_x)] = kFirstNone;
  }
}
__global__ __launch_bounds__(256) void kseq_stage(const RasterJob * jobs, const SeqStage g)
{
  const RasterJob job = jobs[blockIdx.y];          // (a copy: kseq_links)
  stage_copy(g, job, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// kseq_tile: SmearPoint (Mapper.h:1152-1183) of one job's stamped points, tile by tile, for smear kernels of 8 x 8 .. 41 x 41
// cells -- k_raster_tile_reg laid out for latency: eight waves per tile (8 tile rows each: twice the waves per point list), a
// point's eight row values read without a test (the kernel table has 15 zero rows above and below: a footprint that overlaps
// the wave's band at all stays inside it), the table copied from a ready-made image instead of built per workgroup, and the
// dependent loads in front of the first stamp cut from five to two: (tile, list start, count) in one record, the cell in the
// list entry itself.
constexpr int kTabPitch = 192;               // 64 zeros | <= 41 values | zeros: index 64 + lane - fx is in [1, 167]
constexpr int kTabGuard = 15;                // zero rows above and below
constexpr int kTabRows = 41 + 2 * kTabGuard;
constexpr int kTileStageBlocks = 16;
__global__ __launch_bounds__(512) void kseq_tile(const RasterJob * jobs, const uint4 * __restrict__ tab, int tile_blocks, const SeqStage stage)
{
  const RasterJob job = jobs[blockIdx.y];          // (a copy: kseq_links)
  if ((int)blockIdx.x >= tile_blocks) {
    stage_copy(stage, job, ((int)blockIdx.x - tile_blocks) * 512 + (int)threadIdx.x, ((int)gridDim.x - tile_blocks) * 512);
    return;
  }
  __shared__ __attribute__((aligned(16))) uint8_t s_tab[kTabRows * kTabPitch];
  __shared__ int32_t s_pxy[512];
  const int4 * __restrict__ work2 = reinterpret_cast<const int4 *>(job.work2);
  const int n_work = job.n_work[0];
  if ((int)blockIdx.x >= n_work) {return;}
  const int k = job.kernel_size, hk = k / 2, ws = job.ws, height = job.height, tiles_w = job.tiles_w;
  uint8_t * const grid = job.grid;
  const int32_t * const list = job.list;
  const int tid = threadIdx.x;
  // a workgroup's tiles one after the other: the record of the tile after the next and the next tile's first 512 list entries are on
  // their way while this one is stamped (a batch gives a workgroup several tiles; as load -> load -> stamp per tile the chain of two
  // dependent round trips was what a tile cost)
  const int4 none = make_int4(0, 0, 0, 0);
  int4 wk = work2[blockIdx.x];
  int4 wk1 = (int)blockIdx.x + tile_blocks < n_work ? work2[blockIdx.x + tile_blocks] : none;
  int pk0 = tid < min(512, wk.z) ? list[wk.y + tid] : 0;
  for (int i = tid; i < kTabRows * kTabPitch / 16; i += 512) {reinterpret_cast<uint4 *>(s_tab)[i] = tab[i];}
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int band = 8 * wave;                   // first tile row of this wave
  for (int w = blockIdx.x; w < n_work; w += tile_blocks) {
    const int t = wk.x, begin = wk.y, count = wk.z;
    const int4 wk2 = w + 2 * tile_blocks < n_work ? work2[w + 2 * tile_blocks] : none;
    const int pk1 = tid < min(512, wk1.z) ? list[wk1.y + tid] : 0;
    const int ty = t / tiles_w, tx = t - ty * tiles_w;
    const int ox = tx * kRasterTile, oy = ty * kRasterTile;       // grid cell of the tile's corner
    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {acc[j] = 0u;}
    for (int chunk = 0; chunk < count; chunk += 512) {
      const int here = min(512, count - chunk);
      __syncthreads();                                            // table complete / previous chunk consumed
      if (tid < here) {s_pxy[tid] = chunk == 0 ? pk0 : list[begin + chunk + tid];}
      __syncthreads();
      for (int q0 = 0; q0 < here; q0 += 64) {
        const int cnt = min(64, here - q0);
        const int pk = s_pxy[q0 + lane];                          // (entries beyond `here` are stale: masked out below)
        // which of the 64 points reach this wave's eight rows: every lane tests its own, the wave visits the hits only (a footprint
        // of 19 rows meets a band of 8 in two cases of five; tested one point after the other the misses cost as much as the hits)
        const int dl = (pk >> 16) - hk - oy - band;
        unsigned long long hits = __ballot(lane < cnt && !(dl > 7 || dl + k <= 0));
        while (hits) {
          const int q = __builtin_ctzll(hits);
          hits &= hits - 1ull;
          const int v = __builtin_amdgcn_readlane(pk, q);
          const int fx = (v & 0xffff) - hk - ox, fy = (v >> 16) - hk - oy;      // footprint corner relative to the tile
          const int d = fy - band;                                // footprint row of tile row band + j: j - d
          const uint8_t * row0 = s_tab + (kTabGuard - d) * kTabPitch + 64 - fx + lane;
          uint32_t r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {r[j] = row0[j * kTabPitch];}
#pragma unroll
          for (int j = 0; j < 8; ++j) {acc[j] = max(acc[j], r[j]);}
        }
      }
    }
    wk = wk1; wk1 = wk2; pk0 = pk1;
    // write the band: 8 rows x 64 bytes, lane = column (clipped to the grid)
    const int x = ox + lane;
    if (x < ws) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int y = oy + band + j;
        if (y < height) {grid[(size_t)y * ws + x] = (uint8_t)acc[j];}
      }
    }
  }
}

static SeqStage make_stage(const SeqStageArgs * a)
{
  SeqStage g;
  std::memset(&g, 0, sizeof(g));
  if (a) {
    g.src = reinterpret_cast<const uint4 *>(a->h_stage); g.dst = reinterpret_cast<uint4 *>(a->d_stage); g.units = (int)((a->bytes + 15) / 16);
    g.sums = a->sums; g.n_sums = (int)a->n_sums; g.out = a->out; g.out_words = (int)a->out_words;
  }
  return g;
}

void launch_seq_stage(const RasterJob * d_jobs, int32_t n_jobs, const SeqStageArgs * a, void * stream)
{
  if (n_jobs <= 0) {return;}
  const SeqStage g = make_stage(a);
  const int blocks = a ? std::max(1, std::min(64, (int)((std::max<size_t>(g.units, a->n_sums) + 255) / 256))) : 8;
  hipLaunchKernelGGL(kseq_stage, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs, g);
}

// the padded image of the smear kernel kseq_tile copies into LDS (host memory, kTabRows x kTabPitch bytes)
size_t seq_tile_table_bytes() {return (size_t)kTabRows * kTabPitch;}
void seq_tile_table(const uint8_t * kernel, int32_t kernel_size, uint8_t * out)
{
  std::fill(out, out + seq_tile_table_bytes(), (uint8_t)0);
  for (int row = 0; row < kernel_size; ++row) {
    for (int col = 0; col < kernel_size; ++col) {out[(kTabGuard + row) * kTabPitch + 64 + col] = kernel[row * kernel_size + col];}
  }
}

void launch_seq_tile(const RasterJob * d_jobs, int32_t n_jobs, const uint8_t * d_tab, int32_t max_points, int32_t max_tiles, const SeqStageArgs * a,
  void * stream)
{
  if (n_jobs <= 0) {return;}
  const SeqStage g = make_stage(a);
  // one job: a workgroup per tile (latency).  A batch: 64 workgroups per job, each walking its share of the job's tiles with the next
  // tile's record and list on their way -- one copy of the kernel image into LDS per ~6 tiles instead of one per tile, and no
  // workgroups that find nothing to do (2048 per job, ~370 of them with a tile: 1070 us per 224 jobs; 64: 822; 16 or 256: 880)
  static const int env_blocks = std::getenv("KH_TILE_BLOCKS") ? std::atoi(std::getenv("KH_TILE_BLOCKS")) : 0;
  const int batch_blocks = env_blocks > 0 ? env_blocks : 64;
  const int tile_blocks = std::max(1, std::min(std::min(max_tiles, 4 * max_points), n_jobs > 1 ? batch_blocks : 1024));
  hipLaunchKernelGGL(kseq_tile, dim3(tile_blocks + (a ? kTileStageBlocks : 4), n_jobs), dim3(512), 0, (hipStream_t)stream, d_jobs,
    reinterpret_cast<const uint4 *>(d_tab), tile_blocks, g);
}

// ---------------------------------------------------------------------------------------------
// kseq_score: GridIndexLookup::ComputeOffsets (Karto.h:6844-6894) and GetResponse (Mapper.cpp:1172-1208) of one search on a
// linear lattice, in one launch.  Workgroup = (angle, scoring tile, slice of kSeqSlice beams); its four waves compute the slice's
// table entries (bit-exact, as k_offsets: every wave for itself, 64 beams at a time, lane = beam), wave c then walks the
// beams of alignment class c exactly as k_score does -- aligned dword loads of the window's rows, packed 16-bit sums -- and
// the slice's sums are ADDED to the volume (integers: the order does not matter).  One job has 21 angles: with the beams of
// an angle in one workgroup (K3) the search is 21 workgroups walking 270 beams per wave one after the other; cut into slices
// it is 357 workgroups of 16 per wave, eight beams' rows in flight.  A window of several tiles (the loop-closure search: 81 x 81
// poses two cells apart = 3 x 3 tiles of 31 x 28) has a workgroup per tile and slice, and each tests ITS tile's rectangle against
// the occupancy block map (what k_offsets' per-tile lists do): 126 workgroups walking long lists become 3 213 short ones, and the
// column-decimated copies of the grid are not needed.  Responses, best and ties follow in kseq_cells / kseq_final.
template <int SX, int RY>
__global__ __launch_bounds__(256) void kseq_score(const uint8_t * jobp, int slices, int tiles_x, int tiles_y)
{
  const CorrJob job = *reinterpret_cast<const CorrJob *>(jobp);          // (a copy: kseq_links)
  constexpr int PX = (SX == 1) ? kTileSpan : (kTileSpan + 1) / 2;
  constexpr int TY = 4 * RY;
  const int tiles = tiles_x * tiles_y;
  const int a = blockIdx.x / (slices * tiles), rest = blockIdx.x - a * (slices * tiles);
  const int tile = rest / slices, sl = rest - tile * slices;
  const int x0 = (tile % tiles_x) * PX, y0 = (tile / tiles_x) * TY;      // first pose of the tile
  constexpr int NB = (SX == 1) ? 4 : 2;
  constexpr int UB = 8;                    // beams (UB * RY row loads) in flight per wave
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lx = lane & 15, ly = lane >> 4;
  __shared__ int32_t s_tile[TY * PX];
  __shared__ int32_t s_list[4][64];
  __shared__ int32_t s_slow[kSeqSlice];
  __shared__ int32_t s_nslow;
  for (int i = threadIdx.x; i < TY * PX; i += 256) {s_tile[i] = 0;}
  if (threadIdx.x == 0) {s_nslow = 0;}
  __syncthreads();
  const int P = job.n_points;
  const int b0 = sl * kSeqSlice, b1 = min(P, b0 + kSeqSlice);
  const double cosine = job.cos_sin[2 * a], sine = job.cos_sin[2 * a + 1];
  const int64_t bmin = job.base0;
  const int64_t bmax = (int64_t)job.base0 + (int64_t)(job.nx - 1) * job.sx + (int64_t)(job.ny - 1) * job.sy_ws;
  const int32_t xs = (job.nx - 1) * job.sx + 1;              // cells a window covers in x
  const int ws = job.ws;
  const float inv_ws = 1.0f / (float)ws;
  const int64_t data_size = job.data_size, pad = job.pad;
  const uint32_t * const bmp = job.blockmap;
  const int cls = wave & 3;
  const int s = (cls + x0 * SX) & 3;                         // byte of a class-c window's first dword (of this tile) that belongs to pose x0
  const uint32_t sel = (s & 1) ? 0x0c030c01u : 0x0c020c00u;   // SX == 2: the even or the odd bytes
  const gbyte * gbase = as_global(job.grid) + ((int64_t)job.base0 + x0 * SX - s);
  // the tile's rectangle relative to the window start (cells), for the occupancy test
  const int32_t tx_lo = x0 * job.sx, tx_hi = (min(job.nx, x0 + PX) - 1) * job.sx;
  const int32_t ty_lo = y0 * job.sy_cells, ty_hi = (min(job.ny, y0 + TY) - 1) * job.sy_cells;
  uint32_t voff[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    int yi = y0 + r * 4 + ly;
    yi = yi < job.ny ? yi : job.ny - 1;                       // rows beyond ny are clamped (sums discarded)
    voff[r] = (uint32_t)(4 * lx) + (uint32_t)yi * (uint32_t)job.sy_ws;
  }
  int32_t acc[RY][NB];
  uint32_t lo[RY], hi[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    lo[r] = 0; hi[r] = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {acc[r][b] = 0;}
  }
  // (a slice is at most kSeqSlice = 64 beams x 100 per byte: the packed 16-bit sums cannot overflow, one flush at the end)
  for (int base = b0; base < b1; base += 64) {
    const int i = base + lane;
    int32_t idx = kInvalidScan;
    bool fast = false;
    int mycls = 0;
    if (i < b1) {
      if (!job.invalid[i]) {
        const double lxp = job.local[2 * i], lyp = job.local[2 * i + 1];
        // Karto.h:6879-6887: rotate, add the grid offset, WorldToGrid subtracts it again
        const double ox = cosine * lxp - sine * lyp;
        const double oy = sine * lxp + cosine * lyp;
        const double gxd = ((ox + job.grid_off_x) - job.grid_off_x) * job.scale;
        const double gyd = ((oy + job.grid_off_y) - job.grid_off_y) * job.scale;
        const int32_t gx = d_to_int(d_round(gxd)), gy = d_to_int(d_round(gyd));
        idx = (int32_t)((uint32_t)gx + (uint32_t)gy * (uint32_t)ws);   // base Grid::GridIndex, no ROI
      }
      if (wave == 0 && tile == 0) {job.table[(size_t)a * P + i] = idx;}
      if (idx != kInvalidScan && !((int64_t)idx + bmax < 0 || (int64_t)idx + bmin >= data_size)) {
        if ((int64_t)idx + bmin >= -pad && (int64_t)idx + bmax < data_size + pad) {
          fast = true;
          if (bmp) {
            // a window none of whose 32 x 32 blocks was touched by a stamp adds 0 to every pose (as k_offsets)
            const int32_t start = (int32_t)((int64_t)idx + bmin);
            int32_t wy0 = (int32_t)((float)start * inv_ws);
            int32_t wx0 = start - wy0 * ws;
            while (wx0 < 0) {wx0 += ws; --wy0;}
            while (wx0 >= ws) {wx0 -= ws; ++wy0;}
            if (wx0 + xs <= ws && !window_has_blocks(bmp, job.bm_w, job.bm_h, wx0 + tx_lo, wy0 + ty_lo, wx0 + tx_hi, wy0 + ty_hi, job.bshift)) {fast = false;}
          }
          mycls = (int)(((int64_t)idx + bmin) & 3);
        } else if (wave == 0) {
          s_slow[atomicAdd(&s_nslow, 1)] = idx;               // needs the per-pose range check
        }
      }
    }
    const bool mine_here = fast && mycls == cls;
    const unsigned long long mask = __ballot(mine_here);
    const int cnt = __builtin_popcountll(mask);
    if (mine_here) {s_list[wave][__builtin_popcountll(mask & ((1ull << lane) - 1ull))] = idx;}
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes have landed (only it reads them)
    __builtin_amdgcn_wave_barrier();
    const int32_t mine = lane < cnt ? s_list[wave][lane] : 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    int k = 0;
    for (; k + UB <= cnt; k += UB) {
      uint32_t w[UB][RY];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const gbyte * wb = gbase + __builtin_amdgcn_readlane(mine, k + u);
#pragma unroll
        for (int r = 0; r < RY; ++r) {w[u][r] = *reinterpret_cast<const gu32 *>(wb + voff[r]);}
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
#pragma unroll
        for (int r = 0; r < RY; ++r) {
          if (SX == 1) {
            lo[r] += w[u][r] & 0x00ff00ffu;                                // [0, b2, 0, b0]
            hi[r] += __builtin_amdgcn_perm(0u, w[u][r], 0x0c030c01u);      // [0, b3, 0, b1]
          } else {
            lo[r] += __builtin_amdgcn_perm(0u, w[u][r], sel);
          }
        }
      }
    }
    for (; k < cnt; ++k) {
      const gbyte * wb = gbase + __builtin_amdgcn_readlane(mine, k);
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        const uint32_t w = *reinterpret_cast<const gu32 *>(wb + voff[r]);
        if (SX == 1) {
          lo[r] += w & 0x00ff00ffu;
          hi[r] += __builtin_amdgcn_perm(0u, w, 0x0c030c01u);
        } else {
          lo[r] += __builtin_amdgcn_perm(0u, w, sel);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    if (SX == 1) {
      acc[r][0] = lo[r] & 0xffffu; acc[r][2] = lo[r] >> 16;
      acc[r][1] = hi[r] & 0xffffu; acc[r][3] = hi[r] >> 16;
    } else {
      acc[r][0] = lo[r] & 0xffffu; acc[r][1] = lo[r] >> 16;
    }
  }
  // merge the four waves' partial sums: byte position j of the aligned tile row is pose (j - s) / SX
#pragma unroll
  for (int r = 0; r < RY; ++r) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int j = 4 * lx + ((SX == 1) ? b : 2 * b + (s & 1));
      const int x = (j - s) / SX;
      if (j >= s && x < PX && acc[r][b] != 0) {atomicAdd(&s_tile[(r * 4 + ly) * PX + x], acc[r][b]);}
    }
  }
  __syncthreads();
  const int n_slow = s_nslow;
  const size_t plane = (size_t)job.nx * job.ny;
  for (int p = threadIdx.x; p < TY * PX; p += 256) {
    const int xi = x0 + p % PX, yi = y0 + p / PX;
    if (xi >= job.nx || yi >= job.ny) {continue;}
    int32_t sum = s_tile[p];
    if (n_slow > 0) {
      // per-pose range check exactly as GetResponse does it (Mapper.cpp:1192-1197)
      const int64_t pose = (int64_t)job.bx[xi] + (int64_t)job.by[yi];
      for (int j = 0; j < n_slow; ++j) {
        const int64_t at = pose + s_slow[j];
        if (at >= 0 && at < data_size) {sum += job.grid[at];}
      }
    }
    if (sum != 0) {atomicAdd(&job.sums[(size_t)a * plane + (size_t)yi * job.nx + xi], sum);}
  }
}

void launch_seq_score(const uint8_t * d_job, int32_t na, int32_t n_points, int32_t nx, int32_t ny, int32_t sx, int32_t ry, void * stream)
{
  const int slices = (n_points + kSeqSlice - 1) / kSeqSlice;
  const int px = sx == 2 ? (kTileSpan + 1) / 2 : kTileSpan;
  const int tiles_x = (nx + px - 1) / px, tiles_y = (ny + 4 * ry - 1) / (4 * ry);
  const dim3 grid((unsigned int)(na * slices * tiles_x * tiles_y));
  hipStream_t s = (hipStream_t)stream;
#define KH_SEQ_SCORE(SXV, RYV) hipLaunchKernelGGL((kseq_score<SXV, RYV>), grid, dim3(256), 0, s, d_job, slices, tiles_x, tiles_y)
  if (sx == 2) {
    if (ry == 8) {KH_SEQ_SCORE(2, 8);} else if (ry == 7) {KH_SEQ_SCORE(2, 7);} else if (ry == 4) {KH_SEQ_SCORE(2, 4);} else {KH_SEQ_SCORE(2, 1);}
  } else {
    if (ry == 8) {KH_SEQ_SCORE(1, 8);} else if (ry == 7) {KH_SEQ_SCORE(1, 7);} else if (ry == 4) {KH_SEQ_SCORE(1, 4);} else {KH_SEQ_SCORE(1, 1);}
  }
#undef KH_SEQ_SCORE
}

// ---------------------------------------------------------------------------------------------
// kseq_cells: the search-space probabilities (per-cell maxima, cell_maxima) into the device's result block and the host's, and
// the best response of the search (Mapper.cpp:775-800): the largest cell maximum.
__global__ __launch_bounds__(256) void kseq_cells(const uint8_t * jobp, unsigned long long * h_lattice)
{
  const CorrJob job = *reinterpret_cast<const CorrJob *>(jobp);          // (a copy: kseq_links)
  double best = cell_maxima(job, (int)blockIdx.x, (int)gridDim.x, h_lattice);
  if (threadIdx.x < 64) {
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      const double o = __shfl_xor(best, sft);
      best = o > best ? o : best;
    }
    if (threadIdx.x == 0 && best > 0.0) {atomicMax(&job.out[0], (unsigned long long)__double_as_longlong(best));}
  }
}

void launch_seq_cells(const uint8_t * d_job, int32_t plane, unsigned long long * h_lattice, void * stream)
{
  const int blocks = std::max(1, std::min(1024, (plane + 63) / 64));
  hipLaunchKernelGGL(kseq_cells, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_job, h_lattice);
}

// ---------------------------------------------------------------------------------------------
// inclusive prefix sum over the lanes of a wave in registers (DPP row shifts + the two row broadcasts): lane 63 holds the total
__device__ __forceinline__ int wave_prefix_add(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
  return v;
}

// The end of a match in three short launches (a kernel boundary costs 1.5 us; ONE workgroup pulling the fine pass's 107 000
// scattered grid bytes through its compute unit's miss queue cost 58):
//   kseq_ties  ONE workgroup: the poses within KT_TOLERANCE of the best (Mapper.cpp:802-817) -- only cells whose maximum ties
//              with the best can hold one; their (cell, angle) pairs are dealt over the threads.  When the coarse pass has exactly
//              ONE best pose -- the common case -- its average is that pose: x and y are lattice values (centre + offset, the
//              host's doubles), the heading atan2(sin h, cos h) comes from a table the host made per search angle; the fine
//              search's lattice indices (Mapper.cpp:649-662) follow by WorldToGrid's IEEE operations.
//   kseq_fine  the fine pass around it (Mapper.cpp:621-629: 3 x 3 cells, naf angles), a WAVE per (angle, 64 beams): the angles'
//              cosines and sines from the host's table for coarse angle a (no libm on the device), every lookup with
//              GetResponse's range check, wave-level sums, nine atomics per wave.
//   kseq_done  ONE workgroup: responses, best, ties of the fine pass; everything the host needs into host-coherent memory, then
//              the flag.
// The host checks the centre and the lattice indices the device used against its own and redoes the fine pass itself if they
// differ or if the coarse pass had several best poses (their mean needs atan2).
__global__ __launch_bounds__(1024) void kseq_ties(const SeqFinalArgs A)
{
  __shared__ int s_nt, s_ncell;
  __shared__ uint32_t s_tie0;
  __shared__ int32_t s_cells[1024];
  const int tid = threadIdx.x;
  // everything read from the job, once (behind a store the compiler must assume the job block itself changed)
  const CorrJob jr = *reinterpret_cast<const CorrJob *>(A.job);          // (a copy: kseq_links)
  const int nx = jr.nx, na = jr.na, plane = jr.nx * jr.ny, ws = jr.ws;
  const bool penal = jr.do_penalize != 0;
  const double denom = jr.denom, goff_x = jr.grid_off_x, goff_y = jr.grid_off_y, scale = jr.scale;
  const int32_t * const sums = jr.sums;
  unsigned long long * const out = jr.out;
  const double * const dist_pen = jr.dist_pen, * const ang_pen = jr.ang_pen;
  if (A.dbg && tid == 0) {A.dbg[0] = (long long)wall_clock64();}
  const double best = __longlong_as_double((long long)out[0]);
  const unsigned long long * lattice = out + kOutHeaderWords;
  uint32_t * tie_idx = reinterpret_cast<uint32_t *>(out + 2);
  uint32_t * h_tie = reinterpret_cast<uint32_t *>(A.h_out + 2);
  if (tid == 0) {s_nt = 0; s_tie0 = 0u;}
  for (int i = tid; i < A.naf * 9; i += 1024) {A.fsum[i] = 0;}
  for (int cell0 = 0; cell0 < plane; cell0 += 1024) {
    if (tid == 0) {s_ncell = 0;}
    __syncthreads();
    const int cell = cell0 + tid;
    if (cell < plane) {
      const double dm = __longlong_as_double((long long)lattice[cell]) - best;
      if (dm < 0.0 ? dm >= -1e-06 : dm <= 1e-06) {s_cells[atomicAdd(&s_ncell, 1)] = cell;}
    }
    __syncthreads();
    const int pairs = s_ncell * na;
    for (int pr = tid; pr < pairs; pr += 1024) {
      const int c = s_cells[pr / na], a = pr % na;
      // pose_response (Mapper.cpp:1204, 671-685)
      double response = (double)sums[(size_t)a * plane + c] / denom;
      if (penal) {
        const double d0 = response - 0.0;
        if (!(d0 < 0.0 ? d0 >= -1e-06 : d0 <= 1e-06)) {response *= (dist_pen[c] * ang_pen[a]);}
      }
      const double delta = response - best;
      if (delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06) {
        const int slot = atomicAdd(&s_nt, 1);
        const uint32_t idx = (uint32_t)((size_t)c * na + a);          // (y * nX + x) * nA + a
        if (slot < kTieCap) {tie_idx[slot] = idx; h_tie[slot] = idx;}
        if (slot == 0) {s_tie0 = idx;}
      }
    }
  }
  __syncthreads();
  if (tid != 0) {return;}
  const int n_ties = s_nt;
  out[1] = (unsigned long long)n_ties;
  A.h_out[0] = out[0]; A.h_out[1] = (unsigned long long)n_ties;
  const bool fine = A.refine != 0 && n_ties == 1;
  SeqMid mid;
  mid.fine = fine ? 1 : 0; mid.a = 0; mid.pad = 0;
  for (int k = 0; k < 3; ++k) {mid.bx[k] = 0; mid.by[k] = 0; mid.centre[k] = 0.0;}
  if (fine) {
    const uint32_t t = s_tie0;
    const int a = (int)(t % (uint32_t)na);
    const uint32_t xy = t / (uint32_t)na;
    const int xi = (int)(xy % (uint32_t)nx), yi = (int)(xy / (uint32_t)nx);
    // the mean of one pose, the way the host takes it (sum from zero, divided by the count)
    double ax = 0.0, ay = 0.0;
    ax += A.cx + A.xp[xi];
    ay += A.cy + A.yp[yi];
    const int32_t count = 1;
    ax /= count; ay /= count;
    mid.a = a; mid.centre[0] = ax; mid.centre[1] = ay; mid.centre[2] = A.heading[a];
    // lattice base indices of the fine search: operator()(y), Mapper.cpp:649-662
    for (int k = 0; k < 3; ++k) {
      const double newPositionX = ax + A.fxp[k];
      const double gx = (newPositionX - goff_x) * scale;
      mid.bx[k] = d_to_int(d_round(gx)) + A.roi_x;
      const double newPositionY = ay + A.fyp[k];
      const double gy = (newPositionY - goff_y) * scale;
      mid.by[k] = (d_to_int(d_round(gy)) + A.roi_y) * ws;
    }
    A.h_fine->a = a; A.h_fine->xi = xi; A.h_fine->yi = yi;
    for (int k = 0; k < 3; ++k) {A.h_fine->centre[k] = mid.centre[k]; A.h_fine->bx[k] = mid.bx[k]; A.h_fine->by[k] = mid.by[k];}
  }
  *A.mid = mid;
  if (A.dbg) {A.dbg[1] = (long long)wall_clock64();}
}

__global__ __launch_bounds__(64) void kseq_fine(const SeqFinalArgs A, int slices)
{
  const SeqMid & mid = *A.mid;
  if (mid.fine == 0) {return;}
  const CorrJob job = *reinterpret_cast<const CorrJob *>(A.job);          // (a copy: kseq_links)
  const int lane = threadIdx.x;
  const int k = blockIdx.x / slices, i = (blockIdx.x - k * slices) * 64 + lane;
  const int P = job.n_points, naf = A.naf;
  const bool on = i < P;
  const double * cs = A.fine_cos_sin + ((size_t)mid.a * naf + k) * 2;
  const double cosine = cs[0], sine = cs[1];
  const int64_t data_size = job.data_size;
  int32_t idx = kInvalidScan;
  if (on && !job.invalid[i]) {
    const double lxp = job.local[2 * i], lyp = job.local[2 * i + 1];
    // Karto.h:6879-6887: rotate, add the grid offset, WorldToGrid subtracts it again
    const double ox = cosine * lxp - sine * lyp;
    const double oy = sine * lxp + cosine * lyp;
    const double gxd = ((ox + job.grid_off_x) - job.grid_off_x) * job.scale;
    const double gyd = ((oy + job.grid_off_y) - job.grid_off_y) * job.scale;
    const int32_t gx = d_to_int(d_round(gxd)), gy = d_to_int(d_round(gyd));
    idx = (int32_t)((uint32_t)gx + (uint32_t)gy * (uint32_t)job.ws);
  }
  if (on) {A.fine_table[(size_t)k * P + i] = idx;}
  const gbyte * grid = as_global(job.grid);
  uint32_t v[9];
  uint32_t ok = 0u;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int64_t at = (int64_t)mid.by[j / 3] + mid.bx[j % 3] + idx;
    const bool in = at >= 0 && at < data_size;                 // Mapper.cpp:1192-1197
    ok |= in ? (1u << j) : 0u;
    v[j] = grid[in ? (uint32_t)at : 0u];
  }
  if (!on || idx == kInvalidScan) {ok = 0u;}                   // Mapper.cpp:1194
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int sum = wave_prefix_add(((ok >> j) & 1u) ? (int)v[j] : 0);
    if (lane == 63 && sum != 0) {atomicAdd(&A.fsum[k * 9 + j], sum);}
  }
}

__global__ __launch_bounds__(1024) void kseq_done(const SeqFinalArgs A)
{
  __shared__ int s_fnt;
  __shared__ unsigned long long s_fbest;
  const int tid = threadIdx.x;
  const SeqMid mid = *A.mid;
  const bool fine = mid.fine != 0;
  const int naf = A.naf;
  const double denom = reinterpret_cast<const CorrJob *>(A.job)->denom;
  if (A.dbg && tid == 0) {A.dbg[2] = (long long)wall_clock64();}
  if (tid == 0) {s_fnt = 0; s_fbest = 0ull;}
  __syncthreads();
  double response = -1.0;
  int fk = 0, fj = 0;
  if (fine && tid < naf * 9) {
    fk = tid / 9; fj = tid - 9 * fk;
    const int32_t sum = A.fsum[tid];
    A.fine_sums[tid] = sum;                                    // [a][y][x] with a 3 x 3 plane
    A.h_fine->sums[tid] = sum;
    response = (double)sum / denom;                            // Mapper.cpp:1204
    if (A.fine_penalize) {
      const double delta = response - 0.0;
      const bool is_zero = delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06;
      if (!is_zero) {response *= (A.fine_dist_pen[fj] * A.fine_ang_pen[(size_t)mid.a * naf + fk]);}
    }
    atomicMax(&s_fbest, (unsigned long long)__double_as_longlong(response));
  }
  __syncthreads();
  if (fine && tid < naf * 9) {
    const double fbest = __longlong_as_double((long long)s_fbest);
    const double delta = response - fbest;
    if (delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06) {
      const int slot = atomicAdd(&s_fnt, 1);
      reinterpret_cast<uint32_t *>(A.h_fine->out + 2)[slot] = (uint32_t)(fj * naf + fk);      // (y * nx + x) * na + a
    }
  }
  // every wave's stores have reached the L2 before the flag's release writes the L2 back
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (fine) {A.h_fine->out[0] = s_fbest; A.h_fine->out[1] = (unsigned long long)s_fnt;}
    A.h_fine->valid = fine ? 1 : 0;
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(A.h_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (A.dbg) {A.dbg[3] = (long long)wall_clock64();}
  }
}

void launch_seq_final(const SeqFinalArgs & args, int32_t n_points, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(kseq_ties, dim3(1), dim3(1024), 0, s, args);
  if (args.refine) {
    const int slices = (n_points + 63) / 64;
    hipLaunchKernelGGL(kseq_fine, dim3(args.naf * slices), dim3(64), 0, s, args, slices);
  }
  hipLaunchKernelGGL(kseq_done, dim3(1), dim3(1024), 0, s, args);
}

}  // namespace kh
