// gfx950 (MI355X / CDNA4) kernels of the karto correlative scan matcher.
//
//   k_raster_* K1  AddScans/AddScan/SmearPoint      Mapper.cpp:1032-1105, Mapper.h:1152-1183 (clear, bin, scan, fill, tile)
//   k_offsets  K2  GridIndexLookup::ComputeOffsets  Karto.h:6844-6894 (+ per-lattice compaction, empty-window skipping)
//   k_score    K3  operator()(y) + GetResponse      Mapper.cpp:641-694, 1172-1208
//   k_ties     K4  best-response tie collection     Mapper.cpp:802-817
//
// Build with -ffp-contract=off: the grid-index roundings and the penalty product must see the
// same IEEE operations, unfused, as the reference's generic x86-64 build.
//
// Scoring design (K3).  For a fixed search angle a and beam i every pose (x, y) of the search
// lattice reads grid[base(x, y) + off[a][i]], and base is linear in (x, y): the nX x nY poses read
// one contiguous window of the grid per beam.  So a wave owns a 64-byte x 4*RY-row window
// (16 lanes x dword across, 4 lanes down, RY rows per lane), walks the beam list once with the
// offset in an SGPR, issues ONE aligned dword load per lane per row (4 lookups) and accumulates
// the four bytes in two packed 2x16-bit registers -- no per-lookup address math, no byte loads.
// The 4 waves of a workgroup take the beams of one alignment class each ((window start) & 3, see
// k_score) and merge through LDS.  The arithmetic is exact integer; the FP64 penalty is applied
// once per pose in the epilogue.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include "kh_internal.hpp"
#include "matcher_device.hpp"

namespace kh
{

// ---------------------------------------------------------------------------------------------
// K0: FindValidPoints (Mapper.cpp:1113-1164) for every (job, base scan) pair of a batch, one LANE per pair.  The walk is
// a sequential state machine (each 0.1 m trigger depends on the previous trigger's point), but the pairs are independent:
// a loop-closure batch holds thousands of them.  Plain IEEE double operations in the reference's order (no libm), so the
// kept set is bit-identical to the host's.  Pass 1 leaves a code per reading (0 no trigger, 1 trigger on the wrong side,
// 2 trigger on the viewpoint's side); the run [previous trigger, this trigger) is emitted iff the trigger's side test
// passes, and the tail after the last trigger never is, so pass 2 walks back and hands every reading the verdict of the
// first trigger behind it.
constexpr int kCodeWords = 128;      // 2-bit codes of up to 2048 readings per lane live in LDS between the two passes
__global__ __launch_bounds__(64) void k_find_valid_lane(const RasterJob * jobs, const ValidItem * items, int n_items)
{
  __shared__ uint32_t s_codes[kCodeWords][64];           // [word][lane]: conflict-free
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_items) {return;}
  const int lane = threadIdx.x;
  const RasterJob & job = jobs[items[t].job];
  const int k = items[t].scan;
  const int n = job.scan_prefix[k + 1] - job.scan_prefix[k];
  const double2 * pts = reinterpret_cast<const double2 *>(job.scan_ptr[k]);
  uint8_t * out = job.active + job.scan_prefix[k];
  const bool in_lds = n <= 16 * kCodeWords;              // longer scans keep their codes in `out` (global) instead
  const double vx = job.view_x, vy = job.view_y;
  const double min_square_distance = 0.1 * 0.1;          // math::Square(0.1), folded in double like the host does
  double fx = 0.0, fy = 0.0;
  bool first_time = true;
  constexpr int kAhead = 8;                              // readings fetched before the state machine consumes them
  uint32_t word = 0;
  for (int base = 0; base < n; base += kAhead) {
    double2 c[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {c[u] = pts[min(base + u, n - 1)];}
#pragma unroll
    for (int u = 0; u < kAhead; ++u) {
      const int it = base + u;
      if (it >= n) {break;}
      const double cx = c[u].x, cy = c[u].y;
      if (first_time && !isnan(cx) && !isnan(cy)) {fx = cx; fy = cy; first_time = false;}
      const double dx = fx - cx, dy = fy - cy;
      uint32_t code = 0;
      if (dx * dx + dy * dy > min_square_distance) {
        const double a = vy - fy;
        const double b = fx - vx;
        const double cc = fy * vx - fx * vy;
        const double ss = cx * a + cy * b + cc;
        fx = cx; fy = cy;
        code = ss < 0.0 ? 1u : 2u;
      }
      if (in_lds) {
        word |= code << (2 * (it & 15));
        if ((it & 15) == 15) {s_codes[it >> 4][lane] = word; word = 0;}
      } else {
        out[it] = (uint8_t)code;
      }
    }
  }
  if (in_lds && (n & 15) != 0) {s_codes[n >> 4][lane] = word;}
  uint32_t carry = 0;
  if (in_lds) {
    for (int w = (n - 1) >> 4; w >= 0; --w) {
      const uint32_t codes = s_codes[w][lane];
#pragma unroll
      for (int b = 15; b >= 0; --b) {
        const int it = 16 * w + b;
        if (it < n) {
          const uint32_t code = (codes >> (2 * b)) & 3u;
          out[it] = (uint8_t)carry;
          if (code) {carry = code == 2u ? 1u : 0u;}
        }
      }
    }
  } else {
    for (int it = n - 1; it >= 0; --it) {
      const uint8_t code = out[it];
      out[it] = (uint8_t)carry;
      if (code) {carry = code == 2 ? 1u : 0u;}
    }
  }
}

__global__ __launch_bounds__(256) void k_find_valid_par(const RasterJob * jobs, const ValidItem * items, int n_items, int max_n)
{
  // one workgroup per scan: a lane owns every 256th reading, so the divergent forward scans of next() cost a lane four or
  // five readings' worth of its slowest neighbour instead of seventeen (one wave per scan: 56 us for 20 scans)
  extern __shared__ double2 s_fv[];
  const int t = blockIdx.x;
  const RasterJob & job = jobs[items[t].job];
  const int k = items[t].scan;
  const int n = job.scan_prefix[k + 1] - job.scan_prefix[k];
  uint8_t * flags = nullptr;
  find_valid_scan<false>(reinterpret_cast<const double2 *>(job.scan_ptr[k]), n, job.active + job.scan_prefix[k], job.view_x, job.view_y, max_n, s_fv, flags);
}

void launch_find_valid(const RasterJob * d_jobs, const ValidItem * d_items, int32_t n_items, int32_t max_n, void * stream)
{
  if (n_items <= 0) {return;}
  // one lane per scan only when a scan's working set would not fit the LDS (> 2048 readings): 6400 scans of the
  // loop-closure batch take 0.29 ms lane-per-scan (every lane busy, one memory latency per reading), 0.6 ms with a wave
  // hopping from trigger to trigger, ~0.13 ms with the data-parallel workgroup per scan
  if (max_n > 2048) {
    hipLaunchKernelGGL(k_find_valid_lane, dim3((n_items + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_jobs, d_items, (int)n_items);
    return;
  }
  const size_t stride_i = (size_t)max_n + 64;
  const size_t lds = (size_t)max_n * sizeof(double2) + 3 * stride_i * sizeof(int32_t) + 2 * stride_i;
  static std::atomic<unsigned long long> attr2_done{0};
  allow_dynamic_lds(reinterpret_cast<const void *>(k_find_valid_par), 64 * 1024, attr2_done);
  hipLaunchKernelGGL(k_find_valid_par, dim3(n_items), dim3(256), lds, (hipStream_t)stream, d_jobs, d_items, (int)n_items, (int)max_n);
}

// ---------------------------------------------------------------------------------------------
// K1: AddScans after Grid::Clear.  The stamp is a commutative byte-wise max (SmearPoint, Mapper.h:1152-1183),
// so the parallel result equals the sequential one; the only order-dependent part of AddScan ("skip if the
// cell is already 100", Mapper.cpp:1093-1096) matters when the kernel holds 100 off-centre and is resolved by
// k_cell_first / k_active_set below.  Tiled so that the read-modify-write traffic stays in LDS:
//
//   k_raster_bin    thread per point: WorldToGrid, ROI test, duplicate test (a second point in the same cell
//                   stamps the same footprint: dropped), occupancy-block marks, incidence counts of the
//                   <= 2 x 2 tiles of 64 x 64 cells the k x k footprint overlaps (k <= 41)
//   k_raster_scan   workgroup per job: exclusive scan of the counts of the non-empty tiles (listed by k_raster_bin) -> list starts
//   k_raster_fill   thread per point: point index into the lists of its tiles
//   k_raster_tile   workgroup per non-empty tile: max-stamps its points into an LDS tile (ds_max_u32 per
//                   cell), then writes the tile's bytes once, coalesced.  HBM sees the grid clear, one
//                   4 KB write per touched tile and the point records -- not k^2 atomics per point.
// Grid::Clear (Karto.h:4612-4615) of every job of the batch in ONE launch: the grid, its occupancy block map and
// the zero-initialised part of the tile scratch (a memset per buffer per job costs more in launches than in bytes).
__global__ __launch_bounds__(256) void k_raster_clear(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  // Only the tiles the previous rasterisation of this slot wrote can hold anything but 0 (k_raster_tile writes whole tiles,
  // k_raster_bin's centre bytes lie in listed tiles, k_tiles_keep records the list): zeroing those is Grid::Clear.  A 4096 x
  // 4093 grid of the sequential preset is 16.8 MB, the ~1000 tiles a chain of scans touches are 4 MB.
  {
    const int n_prev = job.prev_work[0];
    for (int w = blockIdx.x; w < n_prev; w += gridDim.x) {
      const int t = job.prev_work[4 + w];
      const int ty = t / job.tiles_w, tx = t - ty * job.tiles_w;
      const int ox = tx * kRasterTile, oy = ty * kRasterTile;
      // 64 rows x 64 bytes: 256 threads x 16 bytes (ws is a multiple of 8, the tile starts on a multiple of 64)
      const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
      const int y = oy + row, x = ox + 16 * part;
      if (y < job.height && x < job.ws) {
        uint8_t * at = job.grid + (size_t)y * job.ws + x;
        if (x + 16 <= job.ws) {
          *reinterpret_cast<uint2 *>(at) = make_uint2(0u, 0u);
          *reinterpret_cast<uint2 *>(at + 8) = make_uint2(0u, 0u);
        } else {
          for (int i = 0; x + i < job.ws; i += 8) {*reinterpret_cast<uint2 *>(at + i) = make_uint2(0u, 0u);}
        }
      }
    }
  }
  const size_t bm = (size_t)job.bm_w * job.bm_h;
  for (size_t i = tid; i < bm; i += nth) {job.blockmap[i] = 0u;}
  // tile_count | tile_cursor | n_work (+3 pad) are contiguous
  const size_t nz = 2 * (size_t)job.tiles_w * job.tiles_h + 4;
  for (size_t i = tid; i < nz; i += nth) {job.tile_count[i] = 0;}
  if (job.n_foot > 0) {
    for (size_t i = tid; i < (size_t)job.hcap; i += nth) {job.hkeys[i] = kHashEmpty; job.hvals[i] = INT32_MAX;}
  }
}

void launch_raster_clear(const RasterJob * d_jobs, int32_t n_jobs, void * stream)
{
  if (n_jobs <= 0) {return;}
  hipLaunchKernelGGL(k_raster_clear, dim3(512, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs);
}

__device__ __forceinline__ uint32_t hash_slot(const RasterJob & job, uint32_t key)
{
  return (key * 0x9E3779B1u) & (uint32_t)(job.hcap - 1);
}
// slot of `key` in the job's cell table, -1 when the cell holds no valid point
__device__ __forceinline__ int32_t hash_find(const RasterJob & job, uint32_t key)
{
  uint32_t h = hash_slot(job, key);
  for (;;) {
    const uint32_t k = job.hkeys[h];
    if (k == key) {return (int32_t)h;}
    if (k == kHashEmpty) {return -1;}
    h = (h + 1) & (uint32_t)(job.hcap - 1);
  }
}

// Order-dependent rule, step 1 (only when the kernel holds 100 off-centre): the FIRST valid point of every ROI cell.
// Whatever happens to it, the cell is 100 once it has been visited (either it was 100 already, or the point sets it),
// so every later point of the cell is skipped; only the firsts are candidates for stamping.
constexpr int kCellPoints = 4;          // points per thread, loads and table probes of all of them in flight (see k_raster_bin)
__global__ __launch_bounds__(256) void k_cell_first(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  if (job.n_foot <= 0 || (int)blockIdx.x * 256 * kCellPoints >= job.n_points) {return;}
  int pe[kCellPoints];
  bool on[kCellPoints];
  uint8_t act[kCellPoints];
#pragma unroll
  for (int e = 0; e < kCellPoints; ++e) {
    pe[e] = ((int)blockIdx.x * kCellPoints + e) * 256 + (int)threadIdx.x;
    on[e] = pe[e] < job.n_points;
    act[e] = on[e] ? job.active[pe[e]] : (uint8_t)0;
  }
  double2 pt[kCellPoints];
#pragma unroll
  for (int e = 0; e < kCellPoints; ++e) {
    on[e] = on[e] && act[e] != 0;
    pt[e] = on[e] ? job_point(job, pe[e]) : make_double2(0.0, 0.0);
  }
  uint32_t key[kCellPoints], h[kCellPoints];
#pragma unroll
  for (int e = 0; e < kCellPoints; ++e) {
    int32_t gx = 0, gy = 0;
    if (on[e]) {on[e] = roi_cell(job, pt[e], gx, gy);}
    key[e] = (uint32_t)gy * (uint32_t)job.roi_w + (uint32_t)gx;
    h[e] = hash_slot(job, key[e]);
  }
  // probe round by round: the compare-and-swaps of the points still looking for their slot are issued together
  bool any = true;
  while (any) {
    uint32_t seen[kCellPoints];
#pragma unroll
    for (int e = 0; e < kCellPoints; ++e) {seen[e] = on[e] ? atomicCAS(&job.hkeys[h[e]], kHashEmpty, key[e]) : 0u;}
    any = false;
#pragma unroll
    for (int e = 0; e < kCellPoints; ++e) {
      if (on[e]) {
        if (seen[e] == kHashEmpty || seen[e] == key[e]) {atomicMin(&job.hvals[h[e]], pe[e]); on[e] = false;}
        else {h[e] = (h[e] + 1) & (uint32_t)(job.hcap - 1); any = true;}
      }
    }
  }
}

// Step 2a (grid-wide, thread per table slot): every occupied slot becomes a candidate; the slots of the cells in its
// 100-footprint that hold an EARLIER first point are looked up once (later ones can never block it: -1).
__global__ __launch_bounds__(256) void k_cell_links(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (job.n_foot <= 0) {return;}
  const uint32_t key = h < job.hcap ? job.hkeys[h] : kHashEmpty;
  const bool mine_is_candidate = key != kHashEmpty;
  if (mine_is_candidate) {
    const int32_t mine = job.hvals[h];
    const int32_t gy = (int32_t)(key / (uint32_t)job.roi_w), gx = (int32_t)(key - (uint32_t)gy * (uint32_t)job.roi_w);
    // the <= 4 neighbour cells are looked up side by side: every probing round loads the keys of all lookups still running
    int32_t nbr[kMaxFootprint];
    uint32_t want[kMaxFootprint], at[kMaxFootprint];
    bool run[kMaxFootprint];
#pragma unroll
    for (int f = 0; f < kMaxFootprint; ++f) {
      nbr[f] = -1; run[f] = false; want[f] = 0; at[f] = 0;
      if (f < job.n_foot) {
        const int32_t nx = gx + job.foot_dx[f], ny = gy + job.foot_dy[f];
        if (nx >= 0 && nx < job.roi_w && ny >= 0 && ny < job.roi_h) {
          want[f] = (uint32_t)ny * (uint32_t)job.roi_w + (uint32_t)nx;
          at[f] = hash_slot(job, want[f]);
          run[f] = true;
        }
      }
    }
    bool any = true;
    while (any) {
      uint32_t k[kMaxFootprint];
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {k[f] = run[f] ? job.hkeys[at[f]] : kHashEmpty;}
      any = false;
#pragma unroll
      for (int f = 0; f < kMaxFootprint; ++f) {
        if (run[f]) {
          if (k[f] == want[f]) {nbr[f] = (int32_t)at[f]; run[f] = false;}
          else if (k[f] == kHashEmpty) {run[f] = false;}
          else {at[f] = (at[f] + 1) & (uint32_t)(job.hcap - 1); any = true;}
        }
      }
    }
    int32_t nv[kMaxFootprint];
#pragma unroll
    for (int f = 0; f < kMaxFootprint; ++f) {nv[f] = nbr[f] >= 0 ? job.hvals[nbr[f]] : 0;}
#pragma unroll
    for (int f = 0; f < kMaxFootprint; ++f) {if (nbr[f] >= 0 && nv[f] > mine) {nbr[f] = -1;}}
    *reinterpret_cast<int4 *>(job.hnbr + (size_t)h * kMaxFootprint) = make_int4(nbr[0], nbr[1], nbr[2], nbr[3]);
    job.hstate[h] = 0;
  }
  // candidate list (`list` is free until k_raster_fill, n_work[1] was zeroed by the clear): one atomic per wave
  const unsigned long long mask = __ballot(mine_is_candidate);
  if (mask == 0) {return;}
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(mask);
  int base = 0;
  if (lane == leader) {base = atomicAdd(&job.n_work[1], __builtin_popcountll(mask));}
  base = __shfl(base, leader);
  if (mine_is_candidate) {job.list[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = h;}
}

// Step 2b: the reference visits the points in order and stamps one iff its cell is not 100 yet, i.e. iff no EARLIER
// STAMPED point has the cell in its 100-footprint (the centre and, for sigma / res >= 9.9875, the four neighbours).
// Over the candidates that is the greedy independent set in visiting order.  A candidate's fate is fixed as soon as all
// its earlier neighbours are decided, decisions never change, so sweeping until nothing is undecided reaches the
// sequential answer whatever the sweep order (measured: 7-20 sweeps on the loop-closure chains).  One workgroup per job.
__global__ __launch_bounds__(1024) void k_active_set(const RasterJob * jobs)
{
  // One workgroup per job; plain loads and stores (the job's state bytes are this workgroup's alone: the CU's L1 is
  // coherent for its own waves across a barrier).  Every sweep decides what it can and writes the cells it could not decide
  // to the other half of the candidate buffer (`list` holds 4 * n_points ints, the candidates at most n_points): dependency
  // chains run along walls in scan order, most cells are decided in the first sweeps, and the later sweeps -- a dozen of
  // them -- touch only what is left instead of the whole list.
  const RasterJob & job = jobs[blockIdx.x];
  if (job.n_foot <= 0) {return;}
  __shared__ int32_t s_left;
  int32_t * src = job.list;
  int32_t * dst = job.list + (job.n_points > 0 ? job.n_points : 1);
  int n = job.n_work[1];
  for (int sweep = 0; sweep < 1 << 20 && n > 0; ++sweep) {
    if (threadIdx.x == 0) {s_left = 0;}
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += blockDim.x) {
      const int i = i0 + threadIdx.x;
      bool undecided = false;
      int h = 0;
      if (i < n) {
        h = src[i];
        if (job.hstate[h] == 0) {
          const int4 nb4 = *reinterpret_cast<const int4 *>(job.hnbr + (size_t)h * kMaxFootprint);
          const int32_t nb[kMaxFootprint] = {nb4.x, nb4.y, nb4.z, nb4.w};
          bool blocked = false, waiting = false;
#pragma unroll
          for (int f = 0; f < kMaxFootprint; ++f) {
            if (nb[f] < 0) {continue;}
            const uint8_t st = job.hstate[nb[f]];
            blocked = blocked || st == 1;
            waiting = waiting || st == 0;
          }
          if (blocked) {
            job.hstate[h] = 2;
          } else if (!waiting) {
            job.hstate[h] = 1;
          } else {
            undecided = true;
          }
        }
      }
      // survivors -> dst, one LDS atomic per wave
      const unsigned long long mask = __ballot(undecided);
      if (mask) {
        const int lane = threadIdx.x & 63;
        int base = 0;
        if (lane == __builtin_ctzll(mask)) {base = atomicAdd(&s_left, __builtin_popcountll(mask));}
        base = __shfl(base, __builtin_ctzll(mask));
        if (undecided) {dst[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = h;}
      }
    }
    __syncthreads();
    n = s_left;
    __syncthreads();
    int32_t * t = src; src = dst; dst = t;
  }
}

void launch_active_set(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_cap, void * stream)
{
  if (n_jobs <= 0 || max_points <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_cell_first, dim3((max_points + 256 * kCellPoints - 1) / (256 * kCellPoints), n_jobs), dim3(256), 0, s, d_jobs);
  hipLaunchKernelGGL(k_cell_links, dim3((max_cap + 255) / 256, n_jobs), dim3(256), 0, s, d_jobs);
  hipLaunchKernelGGL(k_active_set, dim3(n_jobs), dim3(1024), 0, s, d_jobs);
}

// Adds 1 to counters[t] for every lane of the wave with t >= 0; the lanes that name the same counter share ONE atomic
// (consecutive beams land in the same tile: without this a tile's counter takes hundreds of same-address atomics).
// Returns the value the counter had before this lane's own increment.  Every lane of the wave must call it.
__device__ __forceinline__ int wave_counter_add(int32_t * counters, int t)
{
  const int lane = threadIdx.x & 63;
  int before = 0;
  unsigned long long todo = __ballot(t >= 0);
  while (todo) {
    const int leader = __builtin_ctzll(todo);
    const int t0 = __shfl(t, leader);
    const unsigned long long same = __ballot(t == t0) & todo;
    int base = 0;
    if (lane == leader) {base = atomicAdd(&counters[t0], __builtin_popcountll(same));}
    base = __shfl(base, leader);
    if (t == t0) {before = base + __builtin_popcountll(same & ((1ull << lane) - 1ull));}
    todo &= ~same;
  }
  return before;
}

// kBinPoints points per thread: the kernel is a chain of dependent memory round trips per point (active flag -> scan
// pointer -> reading -> returning atomic on a cold grid line -> block map -> tile counter), so what a thread needs is several
// points in flight, not more threads (one point per thread: 0.9 ms per 256-job batch at 0.6 TB/s of HBM traffic)
constexpr int kBinPoints = 4;
__global__ __launch_bounds__(256) void k_raster_bin(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  if ((int)blockIdx.x * 256 * kBinPoints >= job.n_points) {return;}
  int pe[kBinPoints], cx[kBinPoints], cy[kBinPoints];
  bool on[kBinPoints];
  uint8_t act[kBinPoints];
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    pe[e] = ((int)blockIdx.x * kBinPoints + e) * 256 + (int)threadIdx.x;
    on[e] = pe[e] < job.n_points;
    act[e] = on[e] ? job.active[pe[e]] : (uint8_t)0;
    cx[e] = -1; cy[e] = -1;
  }
  double2 pt[kBinPoints];
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    on[e] = on[e] && act[e] != 0;
    pt[e] = on[e] ? job_point(job, pe[e]) : make_double2(0.0, 0.0);
  }
  int32_t gx[kBinPoints], gy[kBinPoints];
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    gx[e] = 0; gy[e] = 0;
    if (on[e]) {on[e] = roi_cell(job, pt[e], gx[e], gy[e]);}
  }
  if (job.n_foot > 0) {
    // the order-dependent rule: only the first valid point of a cell can stamp, and only if k_active_set let it
#pragma unroll
    for (int e = 0; e < kBinPoints; ++e) {
      if (on[e]) {
        const int32_t h = hash_find(job, (uint32_t)gy[e] * (uint32_t)job.roi_w + (uint32_t)gx[e]);
        on[e] = h >= 0 && job.hvals[h] == pe[e] && job.hstate[h] == 1;
      }
    }
  }
  uint32_t old[kBinPoints];
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    old[e] = 0;
    if (on[e]) {
      cx[e] = gx[e] + job.roi_x; cy[e] = gy[e] + job.roi_y;              // CorrelationGrid::GridIndex, Mapper.h:1122-1128
      if (job.n_foot <= 0) {
        // the kernel's centre is 100 = its maximum: write it now and use the byte as the "cell already stamped" flag
        // (with the cell table of the order-dependent rule the first point of a cell is known already -- hvals -- and
        // the tile kernel writes the centre with the rest of the footprint: no read-modify-write of a cold grid line)
        const int32_t index = cx[e] + cy[e] * job.ws;
        uint32_t * word = reinterpret_cast<uint32_t *>(job.grid) + (index >> 2);
        const uint32_t bit = (uint32_t)kOccupied << (8 * (index & 3));
        old[e] = (atomicOr(word, bit) >> (8 * (index & 3))) & 0xffu;
      }
    }
  }
  const int hk = job.kernel_size / 2;
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    on[e] = on[e] && old[e] == 0;
    int32_t * cell = job.cell_xy + 2 * (size_t)pe[e];
    if (pe[e] < job.n_points) {cell[0] = on[e] ? cx[e] : -1; cell[1] = on[e] ? cy[e] : -1;}
    if (on[e]) {
      const int fx0 = (cx[e] - hk) >> job.bshift, fx1 = (cx[e] + hk) >> job.bshift;
      const int fy0 = (cy[e] - hk) >> job.bshift, fy1 = (cy[e] + hk) >> job.bshift;
      for (int by = fy0; by <= fy1; ++by) {
        for (int bx = fx0; bx <= fx1; ++bx) {
          uint32_t * word = job.blockmap + (size_t)by * job.bm_w + (bx >> 5);
          const uint32_t bit = 1u << (bx & 31);
          if ((*word & bit) == 0) {atomicOr(word, bit);}
        }
      }
    }
  }
  // Incidence counts of the <= 2 x 2 tiles the footprint overlaps (k <= 41 < 64) and the point's rank in each tile's list.
  // The workgroup's consecutive readings fall into a few dozen tiles: they are counted in an LDS table first (keys by
  // linear probing in 256 slots, a crowded table sends the incidence to the global counter), then every occupied slot takes ONE global atomic -- all slots
  // at once -- and a point's rank is its tile's base plus its rank inside the workgroup.  (A wave-by-wave aggregation
  // took a chain of returning global atomics per wave, here and again in k_raster_fill: half of a single match's
  // rasterisation time.)
  constexpr int kSlots = 256;                                                   // one per thread: a single init / flush step
  __shared__ int32_t s_key[kSlots], s_cnt[kSlots], s_base[kSlots];
  s_key[threadIdx.x] = -1; s_cnt[threadIdx.x] = 0;
  __syncthreads();
  int slot_of[kBinPoints][4], local[kBinPoints][4];
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    const int tx0 = on[e] ? (cx[e] - hk) / kRasterTile : 0, tx1 = on[e] ? (cx[e] + hk) / kRasterTile : 0;
    const int ty0 = on[e] ? (cy[e] - hk) / kRasterTile : 0, ty1 = on[e] ? (cy[e] + hk) / kRasterTile : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int tx = tx0 + (q & 1), ty = ty0 + (q >> 1);
      const int t = (on[e] && tx <= tx1 && ty <= ty1) ? ty * job.tiles_w + tx : -1;
      slot_of[e][q] = -1; local[e][q] = 0;
      if (t >= 0) {
        int sl = (int)(((uint32_t)t * 2654435761u) >> 24);                      // 8 bits
        int tries = 0;
        for (; tries < 24; ++tries) {
          const int seen = atomicCAS(&s_key[sl], -1, t);
          if (seen == -1 || seen == t) {break;}
          sl = (sl + 1) & (kSlots - 1);
        }
        if (tries < 24) {
          slot_of[e][q] = sl;
          local[e][q] = atomicAdd(&s_cnt[sl], 1);
        } else {
          // table crowded (hundreds of distinct tiles under one workgroup's readings): this incidence goes to the counter itself
          slot_of[e][q] = -2;
          local[e][q] = atomicAdd(&job.tile_count[t], 1);
          if (local[e][q] == 0) {job.work[atomicAdd(job.n_work, 1)] = t;}
        }
      }
    }
  }
  __syncthreads();
  {
    const int t = s_key[threadIdx.x];
    if (t >= 0) {
      const int before = atomicAdd(&job.tile_count[t], s_cnt[threadIdx.x]);
      s_base[threadIdx.x] = before;
      if (before == 0) {job.work[atomicAdd(job.n_work, 1)] = t;}                // the tile's first points list it
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < kBinPoints; ++e) {
    if (pe[e] < job.n_points) {
      int rk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {rk[q] = slot_of[e][q] >= 0 ? s_base[slot_of[e][q]] + local[e][q] : (slot_of[e][q] == -2 ? local[e][q] : -1);}
      *reinterpret_cast<int4 *>(job.rank + 4 * (size_t)pe[e]) = make_int4(rk[0], rk[1], rk[2], rk[3]);
    }
  }
}

__global__ __launch_bounds__(1024) void k_raster_scan(const RasterJob * jobs)
{
  // exclusive scan of the incidence counts over the NON-EMPTY tiles only (k_raster_bin listed them in `work` when
  // their count left zero): a few hundred entries, not the 16 k tiles of a 8087^2 grid
  const RasterJob & job = jobs[blockIdx.x];
  const int n = job.n_work[0];
  const int per = (n + 1023) / 1024;
  const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
  __shared__ int32_t s_sum[1024];
  int32_t sum = 0;
  for (int w = lo; w < hi; ++w) {sum += job.tile_count[job.work[w]];}
  s_sum[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {                 // Hillis-Steele inclusive scan of the per-thread totals
    int32_t a = 0;
    if ((int)threadIdx.x >= d) {a = s_sum[threadIdx.x - d];}
    __syncthreads();
    s_sum[threadIdx.x] += a;
    __syncthreads();
  }
  int32_t run = s_sum[threadIdx.x] - sum;
  for (int w = lo; w < hi; ++w) {
    const int t = job.work[w];
    job.tile_start[t] = run;
    run += job.tile_count[t];
  }
}

__global__ __launch_bounds__(256) void k_raster_fill(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  int cx = -1, cy = -1;
  if (p < job.n_points) {cx = job.cell_xy[2 * (size_t)p]; cy = job.cell_xy[2 * (size_t)p + 1];}
  const bool on = cx >= 0;
  const int hk = job.kernel_size / 2;
  const int tx0 = on ? (cx - hk) / kRasterTile : 0, tx1 = on ? (cx + hk) / kRasterTile : 0;
  const int ty0 = on ? (cy - hk) / kRasterTile : 0, ty1 = on ? (cy + hk) / kRasterTile : 0;
  // the ranks k_raster_bin handed out are the positions: no counters, no atomics here
  int4 r4 = make_int4(-1, -1, -1, -1);
  if (on) {r4 = *reinterpret_cast<const int4 *>(job.rank + 4 * (size_t)p);}
  const int rk[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int tx = tx0 + (q & 1), ty = ty0 + (q >> 1);
    const int t = (on && tx <= tx1 && ty <= ty1) ? ty * job.tiles_w + tx : -1;
    if (t >= 0) {job.list[job.tile_start[t] + rk[q]] = p;}
  }
}

// Workgroups of 256 threads, up to 8 per CU (18 KB of LDS each): a tile is a few dependent memory latencies long (work
// list -> point list -> cells), so what a CU needs is many tiles in flight, not many threads on one tile (1024-thread
// workgroups, 2 per CU, took 14 us per tile of the loop-closure batch).  The tile's points are staged in LDS in
// chunks of 256 with coalesced loads before any wave starts stamping.
__global__ __launch_bounds__(256) void k_raster_tile(const RasterJob * jobs, const uint8_t * __restrict__ kernel)
{
  const RasterJob & job = jobs[blockIdx.y];
  // one block so that the tile starts >= 40 tile rows into the workgroup's LDS: the stamping loop addresses footprint row r
  // as (column base of row 0) + r * 256 in the instruction's offset field, and the base of a footprint that starts above
  // the tile must not be a negative LDS address
  struct Lds
  {
    uint32_t cells[41 * 41];                     // footprint cell c: value << 16 | row << 8 | column
    int32_t px[256], py[256];
    uint32_t pad[(40 * kRasterTile * 4 - 41 * 41 * 4 - 2048) / 4];
    uint32_t tile[kRasterTile * kRasterTile];
  };
  __shared__ Lds lds;
  uint32_t * s_tile = lds.tile; uint32_t * s_cells = lds.cells; int32_t * s_px = lds.px; int32_t * s_py = lds.py;
  const int k = job.kernel_size, hk = k / 2, kk = k * k;
  const int n_work = job.n_work[0];
  if ((int)blockIdx.x >= n_work) {return;}
  for (int i = threadIdx.x; i < kk; i += blockDim.x) {
    const int row = i / k, col = i - row * k;
    s_cells[i] = ((uint32_t)kernel[i] << 16) | ((uint32_t)row << 8) | (uint32_t)col;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int kWaves = 4;
  // (kernels of fewer than 8 x 8 cells: larger ones take k_raster_tile_reg / kseq_tile.)  64 / kk points per wave pass, one
  // footprint cell per lane; the packed per-cell table keeps the inner loop at a handful of instructions per cell.
  const int ppw = max(1, 64 / kk);                        // points per wave pass
  const int sub = lane / kk;                              // which of them this lane works for
  const int c0 = lane - sub * kk;                         // first cell of this lane
  const bool lane_on = sub < ppw;
  __syncthreads();                                         // s_cells is complete
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int t = job.work[w];
    const int ty = t / job.tiles_w, tx = t - ty * job.tiles_w;
    const int ox = tx * kRasterTile, oy = ty * kRasterTile;       // grid cell of the tile's corner
    for (int i = threadIdx.x; i < kRasterTile * kRasterTile; i += blockDim.x) {s_tile[i] = 0;}
    const int begin = job.tile_start[t], count = job.tile_count[t];
    for (int chunk = 0; chunk < count; chunk += 256) {
      const int here = min(256, count - chunk);
      __syncthreads();                                            // s_tile zeroed / previous chunk consumed
      if ((int)threadIdx.x < here) {
        const int p = job.list[begin + chunk + threadIdx.x];
        // footprint corner relative to the tile
        s_px[threadIdx.x] = job.cell_xy[2 * (size_t)p] - hk - ox;
        s_py[threadIdx.x] = job.cell_xy[2 * (size_t)p + 1] - hk - oy;
      }
      __syncthreads();
      for (int q = wave * ppw + sub; q < here; q += kWaves * ppw) {
        if (!lane_on) {break;}
        const int fx = s_px[q], fy = s_py[q];
        for (int c = c0; c < kk; c += 64) {
          const uint32_t e = s_cells[c];
          const int x = fx + (int)(e & 0xffu), y = fy + (int)((e >> 8) & 0xffu);
          // both inside [0, 64): for two's complement ints, (x | y) is in [0, 64) iff both are
          if ((unsigned)(x | y) < (unsigned)kRasterTile && (e >> 16) != 0) {atomicMax(&s_tile[y * kRasterTile + x], e >> 16);}
        }
      }
    }
    __syncthreads();
    // write the tile: 64 rows x 16 words (clipped to the grid)
    uint32_t * words = reinterpret_cast<uint32_t *>(job.grid);
    for (int i = threadIdx.x; i < kRasterTile * kRasterTile / 4; i += blockDim.x) {
      const int row = i >> 4, wcol = i & 15;
      const int y = oy + row, x = ox + 4 * wcol;
      if (y >= job.height || x >= job.ws) {continue;}
      const uint32_t * src = &s_tile[row * kRasterTile + 4 * wcol];
      const uint32_t packed = src[0] | (src[1] << 8) | (src[2] << 16) | (src[3] << 24);
      words[((size_t)y * job.ws + x) >> 2] = packed;
    }
    __syncthreads();
  }
}

// The same for kernels of 8 x 8 to 41 x 41 cells with the TILE IN REGISTERS: wave w owns tile rows 16w .. 16w + 15, lane =
// tile column, acc[j] = cell (16w + j, lane).  Every wave walks all points of the tile; for a point whose footprint starts
// at (fx, fy) relative to the tile the value for tile row y and this lane's column is kernel[y - fy][lane - fx], read from
// a byte table in LDS whose rows are padded with zeros left and right (no column test) and above and below (rows are taken
// in groups of four under one scalar test): per tile row ONE ds_read_u8 and ONE v_max_u32, no atomics -- the LDS-atomic
// form above is bound by ds_max_u32 at about 4 clocks per wave instruction.
__global__ __launch_bounds__(256) void k_raster_tile_reg(const RasterJob * jobs, const uint8_t * __restrict__ kernel)
{
  const RasterJob & job = jobs[blockIdx.y];
  constexpr int kPitch = 192;                  // 64 zeros | <= 41 values | zeros: index 64 + lane - fx is in [1, 167]
  constexpr int kGuard = 15;                   // zero rows above and below
  __shared__ int32_t s_px[256], s_py[256];
  __shared__ uint8_t s_tab[(41 + 2 * kGuard) * kPitch];
  const int k = job.kernel_size, hk = k / 2;
  const int n_work = job.n_work[0];
  if ((int)blockIdx.x >= n_work) {return;}
  for (int i = threadIdx.x; i < (41 + 2 * kGuard) * kPitch / 4; i += blockDim.x) {reinterpret_cast<uint32_t *>(s_tab)[i] = 0u;}
  __syncthreads();
  for (int i = threadIdx.x; i < k * k; i += blockDim.x) {
    const int row = i / k, col = i - row * k;
    s_tab[(kGuard + row) * kPitch + 64 + col] = kernel[i];
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int band = 16 * wave;                  // first tile row of this wave
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int t = job.work[w];
    const int ty = t / job.tiles_w, tx = t - ty * job.tiles_w;
    const int ox = tx * kRasterTile, oy = ty * kRasterTile;       // grid cell of the tile's corner
    uint32_t acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {acc[j] = 0u;}
    const int begin = job.tile_start[t], count = job.tile_count[t];
    for (int chunk = 0; chunk < count; chunk += 256) {
      const int here = min(256, count - chunk);
      __syncthreads();                                            // table complete / previous chunk consumed
      if ((int)threadIdx.x < here) {
        const int p = job.list[begin + chunk + threadIdx.x];
        s_px[threadIdx.x] = job.cell_xy[2 * (size_t)p] - hk - ox;          // footprint corner relative to the tile
        s_py[threadIdx.x] = job.cell_xy[2 * (size_t)p + 1] - hk - oy;
      }
      __syncthreads();
      for (int q0 = 0; q0 < here; q0 += 64) {
        const int cnt = min(64, here - q0);
        const int mx = s_px[q0 + lane], my = s_py[q0 + lane];      // (entries beyond `here` are stale: not read out below)
        for (int q = 0; q < cnt; ++q) {
          const int fx = __builtin_amdgcn_readlane(mx, q), fy = __builtin_amdgcn_readlane(my, q);
          // footprint rows r = band + j - fy in [0, k) <=> j in [j_lo, j_hi]
          const int j_lo = max(0, fy - band), j_hi = min(15, fy - band + k - 1);
          if (j_hi < j_lo) {continue;}
          const uint8_t * row0 = s_tab + (kGuard + band - fy) * kPitch + 64 - fx + lane;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (j_hi >= 4 * g && j_lo <= 4 * g + 3) {
              uint32_t v[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {v[u] = row0[(4 * g + u) * kPitch];}
#pragma unroll
              for (int u = 0; u < 4; ++u) {acc[4 * g + u] = max(acc[4 * g + u], v[u]);}
            }
          }
        }
      }
    }
    // write the band: 16 rows x 64 bytes, lane = column (clipped to the grid)
    const int x = ox + lane;
    if (x < job.ws) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int y = oy + band + j;
        if (y < job.height) {job.grid[(size_t)y * job.ws + x] = (uint8_t)acc[j];}
      }
    }
  }
}

void launch_raster_tiles(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_tiles, const uint8_t * d_kernel, int32_t kernel_size,
  void * stream)
{
  if (n_jobs <= 0 || max_points <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  // non-empty tiles <= 4 per point and <= all tiles; a workgroup walks several when there are more
  static const int tile_blocks = std::getenv("KH_TILE_BLOCKS") ? std::atoi(std::getenv("KH_TILE_BLOCKS")) : 2048;
  int blocks = std::min(std::min(max_tiles, 4 * max_points), tile_blocks);
  // kernels of >= 8 x 8 cells: tile in registers; smaller ones: LDS atomics, a lane per footprint cell
  if (kernel_size >= 8) {
    hipLaunchKernelGGL(k_raster_tile_reg, dim3(blocks, n_jobs), dim3(256), 0, s, d_jobs, d_kernel);
  } else {
    hipLaunchKernelGGL(k_raster_tile, dim3(blocks, n_jobs), dim3(256), 0, s, d_jobs, d_kernel);
  }
}

void launch_raster(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_tiles, const uint8_t * d_kernel, int32_t kernel_size,
  void * stream)
{
  if (n_jobs <= 0 || max_points <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  dim3 per_point((max_points + 255) / 256, n_jobs);
  hipLaunchKernelGGL(k_raster_bin, dim3((max_points + 256 * kBinPoints - 1) / (256 * kBinPoints), n_jobs), dim3(256), 0, s, d_jobs);
  hipLaunchKernelGGL(k_raster_scan, dim3(n_jobs), dim3(1024), 0, s, d_jobs);
  hipLaunchKernelGGL(k_raster_fill, per_point, dim3(256), 0, s, d_jobs);
  launch_raster_tiles(d_jobs, n_jobs, max_points, max_tiles, d_kernel, kernel_size, stream);
}

// Re-pitched copies (CorrJob::grid2): the tiles the previous rasterisation touched are zeroed, the tiles this one touched
// are copied from the grid, and this rasterisation's tile list becomes the "previous" one.  Traffic: 8 KB per touched tile.
// column-decimated copies (RasterJob::copy_kind 2): cells x .. x + 7 (x a multiple of 8) of grid row y, as two dwords
__device__ __forceinline__ void dec_store(const RasterJob & job, int y, int x, uint32_t d0, uint32_t d1)
{
  const uint32_t even = __builtin_amdgcn_perm(d1, d0, 0x06040200u), odd = __builtin_amdgcn_perm(d1, d0, 0x07050301u);
  const size_t cb = (size_t)job.copy_b;
  auto put = [&](size_t at) {
    *reinterpret_cast<uint32_t *>(job.grid2 + at) = even;
    *reinterpret_cast<uint32_t *>(job.grid2 + cb + at) = odd;
    *reinterpret_cast<uint32_t *>(job.grid2 + 2 * cb + 64 + at) = even;
    *reinterpret_cast<uint32_t *>(job.grid2 + 3 * cb + 64 + at) = odd;
  };
  put((size_t)y * job.pitch2 + (x >> 1));
  // the first 64 columns of a row are repeated behind the row above
  // (row -1 is part of the zero rows in front of the copy: a window that starts there runs on into row 0)
  if (x < 128) {put((size_t)((int64_t)(y - 1) * job.pitch2 + (job.ws >> 1) + (x >> 1)));}
}
__device__ __forceinline__ void repitch_tile(const RasterJob & job, int t, bool zero)
{
  const int ty = t / job.tiles_w, tx = t - ty * job.tiles_w;
  const int ox = tx * kRasterTile, oy = ty * kRasterTile;
  if (job.copy_kind == 2) {
    for (int i = threadIdx.x; i < kRasterTile * kRasterTile / 8; i += blockDim.x) {
      const int row = i >> 3, c8 = i & 7;
      const int y = oy + row, x = ox + 8 * c8;
      if (y >= job.height || x >= job.ws) {continue;}
      const uint2 v = zero ? make_uint2(0u, 0u) : *reinterpret_cast<const uint2 *>(job.grid + (size_t)y * job.ws + x);
      dec_store(job, y, x, v.x, v.y);
    }
    return;
  }
  for (int i = threadIdx.x; i < kRasterTile * kRasterTile / 4; i += blockDim.x) {
    const int row = i >> 4, wcol = i & 15;
    const int y = oy + row, x = ox + 4 * wcol;
    if (y >= job.height || x >= job.ws) {continue;}
    const uint32_t v = zero ? 0u : reinterpret_cast<const uint32_t *>(job.grid)[((size_t)y * job.ws + x) >> 2];
    const size_t at = (size_t)y * job.pitch2 + x;
    *reinterpret_cast<uint32_t *>(job.grid2 + at) = v;
    *reinterpret_cast<uint32_t *>(job.grid2 + job.copy_b + 64 + at) = v;
  }
}

__global__ __launch_bounds__(256) void k_repitch(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  if (!job.grid2) {return;}
  const int n_prev = job.prev_work[0], n_work = job.n_work[0];
  for (int w = blockIdx.x; w < n_prev; w += gridDim.x) {repitch_tile(job, job.prev_work[4 + w], true);}
  // a tile in both lists is zeroed and filled by different workgroups: order them through a second launch instead
}
__global__ __launch_bounds__(256) void k_repitch_fill(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.y];
  if (!job.grid2) {return;}
  const int n_work = job.n_work[0];
  for (int w = blockIdx.x; w < n_work; w += gridDim.x) {repitch_tile(job, job.work[w], false);}
}
__global__ __launch_bounds__(256) void k_repitch_keep(const RasterJob * jobs)
{
  const RasterJob & job = jobs[blockIdx.x];
  const int n_work = job.n_work[0];
  for (int w = threadIdx.x; w < n_work; w += blockDim.x) {job.prev_work[4 + w] = job.work[w];}
  if (threadIdx.x == 0) {job.prev_work[0] = n_work;}
}

void launch_repitch(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_tiles, void * stream, bool any_copies)
{
  if (n_jobs <= 0) {return;}
  hipStream_t s = (hipStream_t)stream;
  const int blocks = std::min(max_tiles, 2048);
  if (any_copies) {
    hipLaunchKernelGGL(k_repitch, dim3(blocks, n_jobs), dim3(256), 0, s, d_jobs);
    hipLaunchKernelGGL(k_repitch_fill, dim3(blocks, n_jobs), dim3(256), 0, s, d_jobs);
  }
  // this rasterisation's tile list becomes the "previous" one: the next Grid::Clear of the slot zeroes exactly these tiles
  hipLaunchKernelGGL(k_repitch_keep, dim3(n_jobs), dim3(256), 0, s, d_jobs);
}

// first use of the copies of a slot: the whole grid, row by row (the copies were zero-filled at allocation)
__global__ __launch_bounds__(256) void k_repitch_full(const RasterJob * jobs, int rows)
{
  const RasterJob & job = jobs[0];
  const int words = job.ws / 4;
  if (job.copy_kind == 2) {
    for (int y = blockIdx.x; y < rows; y += gridDim.x) {
      for (int i = threadIdx.x; i < job.ws / 8; i += blockDim.x) {
        const uint2 v = *reinterpret_cast<const uint2 *>(job.grid + (size_t)y * job.ws + 8 * i);
        dec_store(job, y, 8 * i, v.x, v.y);
      }
    }
    return;
  }
  for (int y = blockIdx.x; y < rows; y += gridDim.x) {
    const uint32_t * src = reinterpret_cast<const uint32_t *>(job.grid + (size_t)y * job.ws);
    uint32_t * a = reinterpret_cast<uint32_t *>(job.grid2 + (size_t)y * job.pitch2);
    uint32_t * b = reinterpret_cast<uint32_t *>(job.grid2 + job.copy_b + 64 + (size_t)y * job.pitch2);
    for (int i = threadIdx.x; i < words; i += blockDim.x) {const uint32_t v = src[i]; a[i] = v; b[i] = v;}
  }
}

void launch_repitch_full(const RasterJob * d_job, int32_t rows, void * stream)
{
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_repitch_full, dim3(std::min(rows, 4096)), dim3(256), 0, s, d_job, (int)rows);
  hipLaunchKernelGGL(k_repitch_keep, dim3(1), dim3(256), 0, s, d_job);
}

// ---------------------------------------------------------------------------------------------
// K2: one workgroup per (angle, job).  Bit-exact table + compaction into the lists K3 walks.
__global__ __launch_bounds__(256) void k_offsets(const uint8_t * jobs, size_t stride)
{
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)blockIdx.y * stride);
  const int a = blockIdx.x;
  if (a >= job.na) {return;}
  // the job's result block starts from zero (best response, tie count, probs): every angle's workgroup clears its share
  {
    const int per = (job.out_words + job.na - 1) / job.na;
    const int hi = min(job.out_words, (a + 1) * per);
    for (int i = a * per + threadIdx.x; i < hi; i += blockDim.x) {job.out[i] = 0ull;}
  }
  __shared__ int32_t s_counts[kClasses + 1];
  __shared__ int32_t s_tcounts[kClasses * 32];
  __shared__ int32_t s_tcounts2[kClasses * 32];
  __shared__ uint32_t s_bm[4096];
  __shared__ int4 s_tile_rect[32];       // per scoring tile: first / last cell (x, y) relative to the window start
  if (threadIdx.x < kClasses + 1) {s_counts[threadIdx.x] = 0;}
  if ((int)threadIdx.x < job.list_tiles && job.list_tiles > 1) {
    const int t = threadIdx.x;
    const int px = job.tile_px, ty_rows = 4 * job.ry;
    const int tx = t % job.tiles_x, ty = t / job.tiles_x;
    const int p0 = tx * px, p1 = min(job.nx, p0 + px) - 1;         // poses of the tile
    const int r0 = ty * ty_rows, r1 = min(job.ny, r0 + ty_rows) - 1;
    s_tile_rect[t] = make_int4(p0 * job.sx, r0 * job.sy_cells, p1 * job.sx, r1 * job.sy_cells);
  }
  if (threadIdx.x < kClasses * 32) {s_tcounts[threadIdx.x] = 0; s_tcounts2[threadIdx.x] = 0;}
  const int lt = job.list_tiles;
  // small occupancy maps (coarse grids) are read from LDS: the per-tile tests make ~100 probes per beam
  const uint32_t * bmp = job.blockmap;
  if (bmp && job.bm_w * job.bm_h <= 4096) {
    for (int i = threadIdx.x; i < job.bm_w * job.bm_h; i += blockDim.x) {s_bm[i] = job.blockmap[i];}
    bmp = s_bm;
  }
  __syncthreads();
  const int P = job.n_points;
  const float inv_ws = 1.0f / (float)job.ws;
  const double cosine = job.cos_sin[2 * a], sine = job.cos_sin[2 * a + 1];
  // index range of the lattice (real poses only)
  const int64_t bmin = job.base0;
  const int64_t bmax = (int64_t)job.base0 + (int64_t)(job.nx - 1) * job.sx + (int64_t)(job.ny - 1) * job.sy_ws;
  int32_t * table = job.table + (size_t)a * P;
  int32_t * fast = job.fast + (size_t)a * kClasses * lt * P;
  int32_t * fast2 = job.grid2 ? job.fast2 + (size_t)a * kClasses * lt * P : nullptr;
  int32_t * slow = job.slow + (size_t)a * P;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    int32_t idx;
    if (job.invalid[i]) {
      idx = kInvalidScan;
    } else {
      const double lx = job.local[2 * i], ly = job.local[2 * i + 1];
      // Karto.h:6879-6887: rotate, add the grid offset, WorldToGrid subtracts it again
      const double ox = cosine * lx - sine * ly;
      const double oy = sine * lx + cosine * ly;
      const double gxd = ((ox + job.grid_off_x) - job.grid_off_x) * job.scale;
      const double gyd = ((oy + job.grid_off_y) - job.grid_off_y) * job.scale;
      const int32_t gx = d_to_int(d_round(gxd)), gy = d_to_int(d_round(gyd));
      idx = (int32_t)((uint32_t)gx + (uint32_t)gy * (uint32_t)job.ws);   // base Grid::GridIndex, no ROI
    }
    table[i] = idx;
    if (idx == kInvalidScan) {continue;}   // Mapper.cpp:1194
    if (job.linear) {
      if ((int64_t)idx + bmax < 0 || (int64_t)idx + bmin >= job.data_size) {continue;}  // off the grid for every pose
      if ((int64_t)idx + bmin >= -(int64_t)job.pad && (int64_t)idx + bmax < (int64_t)job.data_size + job.pad) {
        uint32_t tmask = lt > 1 ? (0xffffffffu >> (32 - lt)) : 1u;
        int32_t wx0 = 0, wy0 = 0;
        bool in_row = false;                 // the window does not wrap around the row end
        if (job.blockmap || job.grid2) {
          // the window of this beam: x0 .. x0 + xs - 1 bytes of grid rows y0 .. y0 + ys - 1
          const int32_t start = (int32_t)((int64_t)idx + bmin);
          wy0 = (int32_t)((float)start * inv_ws);       // start / ws, fixed up below (start < 2^31, ws >= 8)
          wx0 = start - wy0 * job.ws;
          while (wx0 < 0) {wx0 += job.ws; --wy0;}
          while (wx0 >= job.ws) {wx0 -= job.ws; ++wy0;}
          in_row = wx0 + (job.nx - 1) * job.sx + 1 <= job.ws;
        }
        if (job.blockmap) {
          // If no stamp footprint overlaps any of the window's 32 x 32 blocks, every byte of it is 0 and the beam adds
          // nothing to any pose of this angle: leave it out (bit-identical sums).  The same test per scoring tile
          // decides which tile lists the beam joins (large windows are rarely empty as a whole, their tiles
          // often are).  Windows that wrap around the row end (beams beyond the range threshold, Appendix
          // A.3) are kept everywhere.
          const int32_t xs = (job.nx - 1) * job.sx + 1, ys = (job.ny - 1) * job.sy_cells + 1;
          if (in_row) {
            auto any_block = [&](int x_lo, int y_lo, int x_hi, int y_hi) {
              const int bx0 = x_lo >> job.bshift, bx1 = x_hi >> job.bshift;
              // rows above and below the array hold nothing
              const int by0 = max(y_lo, 0) >> job.bshift, by1 = min(y_hi >> job.bshift, job.bm_h - 1);
              // bits bx0 .. bx1 of every block row: two adjacent words cover them (the rows carry a padding
              // word).  No early exit: the probes are independent loads.
              const int wi = bx0 >> 5, sh = bx0 & 31, nb = bx1 - bx0 + 1;
              if (nb > 32) {return true;}                       // wider than two words can show: never skipped
              const unsigned long long span = (1ull << nb) - 1ull;
              unsigned long long any = 0;
              for (int by = by0; by <= by1; ++by) {
                const uint32_t * row = bmp + (size_t)by * job.bm_w + wi;
                any |= (((unsigned long long)row[1] << 32) | row[0]) >> sh;
              }
              any &= span;
              return any != 0;
            };
            if (!any_block(wx0, wy0, wx0 + xs - 1, wy0 + ys - 1)) {continue;}
            if (lt > 1) {
              tmask = 0;
              for (int t = 0; t < lt; ++t) {
                const int4 g = s_tile_rect[t];                    // cells relative to the window start
                if (any_block(wx0 + g.x, wy0 + g.y, wx0 + g.z, wy0 + g.w)) {tmask |= 1u << t;}
              }
            }
          }
        }
        // alignment class of the window start: K3 reads class-c windows with aligned dwords
        const int cls = job.dec ? ((wx0 >> 1) & (kClasses - 1)) : (int)(((int64_t)idx + bmin) & (kClasses - 1));
        // dual-copy layout: the aligned 64-byte segment K3 reads starts at wx0 - cls; the copy in which it does not
        // straddle a 128-byte line (pitch2 is a multiple of 128, so every row of the window sits alike)
        const bool dual = fast2 != nullptr && (in_row || job.dec);
        const int px = job.tile_px;
        while (tmask) {
          const int t = __builtin_ctz(tmask);
          tmask &= tmask - 1;
          const int li = cls * lt + t;
          if (dual && job.dec) {
            // column-decimated copies: the tile's segments start in column k = (wx0 >> 1) + x0 of copy (wx0 & 1); behind the
            // row end they continue at the start of the next row (columns < 64 of it are repeated behind this row, so only a
            // tile that STARTS behind the row end moves down a row)
            const int32_t half = job.ws >> 1;
            const int32_t xo = lt > 1 ? (t % job.tiles_x) * px : 0;
            int32_t k0 = wx0 >> 1, row = wy0;
            if (k0 + xo >= half) {k0 -= half; ++row;}
            const int32_t seg = (k0 + xo) & ~3;
            const int32_t idx2 = row * job.pitch2 + k0 + ((wx0 & 1) ? job.copy_b : 0) + ((seg & 127) > 64 ? 2 * job.copy_b + 64 : 0);
            fast2[(size_t)li * P + atomicAdd(&s_tcounts2[li], 1)] = idx2;
          } else if (dual) {
            // the tile's 64-byte segments start at wx0 + x0 * sx, moved back to the dword boundary (K3's s); with one list
            // for all tiles (lt == 1) the lattice is one tile wide (the host sees to it), i.e. x0 = 0
            const int32_t xo = (lt > 1 ? (t % job.tiles_x) * px : 0) * job.sx;
            const int32_t seg = (wx0 + xo) & ~3;
            const int32_t idx2 = wy0 * job.pitch2 + wx0 + ((seg & 127) > 64 ? job.copy_b + 64 : 0);
            fast2[(size_t)li * P + atomicAdd(&s_tcounts2[li], 1)] = idx2;
          } else {
            fast[(size_t)li * P + atomicAdd(&s_tcounts[li], 1)] = idx;
          }
        }
        continue;
      }
    }
    slow[atomicAdd(&s_counts[kClasses], 1)] = idx;
  }
  __syncthreads();
  if (threadIdx.x < kClasses + 1) {job.counts[kCountsPerAngle * a + threadIdx.x] = s_counts[threadIdx.x];}
  if ((int)threadIdx.x < kClasses * lt) {
    job.tcounts[(size_t)a * kClasses * lt + threadIdx.x] = s_tcounts[threadIdx.x];
    if (job.grid2) {job.tcounts2[(size_t)a * kClasses * lt + threadIdx.x] = s_tcounts2[threadIdx.x];}
  }
  if (threadIdx.x == 0 && job.load_counter) {
    // every entry of a list costs K3 `ry` wave-level dword loads in each scoring tile that walks the list
    long long entries = 0;
    for (int li = 0; li < kClasses * lt; ++li) {entries += s_tcounts[li] + s_tcounts2[li];}
    const long long tiles_per_list = lt > 1 ? 1 : (long long)job.tiles_x * job.tiles_y;
    atomicAdd(job.load_counter, (unsigned long long)(entries * tiles_per_list * job.ry));
  }
}

void launch_offsets(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, void * stream)
{
  if (n_jobs <= 0 || max_na <= 0) {return;}
  hipLaunchKernelGGL(k_offsets, dim3(max_na, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs, stride);
}

// K3.  SX = grid cells per lattice step in x (1: fine / full-resolution search, 2: coarse search).
// Tile = 64 aligned grid bytes x 4*RY lattice rows, of which 61 bytes (61 poses at SX=1, 31 at SX=2)
// carry poses whatever the alignment.
//
// Alignment classes.  A window starts at an arbitrary byte, and a wave of *unaligned* dword loads
// costs ~18 L1 (TCP) tag lookups per instruction -- measured: that, not HBM or VALU, bound the first
// version of this kernel.  K2 therefore buckets every angle's beams by the alignment class
// c = (window start) & 3, and wave c of the workgroup walks class c only: all of its loads are then
// ALIGNED dwords (4-5 tag lookups), and the byte position of pose 0 inside the first dword (s) is a
// wave constant that is applied once, in the epilogue, when the four waves' sums are merged in LDS.
//
// Block -> work mapping is XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so
// id & 7 picks the XCD and each XCD walks whole units (a job, or a contiguous angle range of a job
// when there are fewer than 8 jobs).  All angles of a unit then share one L2, where the windows of
// neighbouring angles overlap by ~85 %.
// A workgroup is AW adjacent angles x 4 alignment classes = 4*AW waves.  Adjacent angles read windows
// that overlap by ~85 % and sweep the beam list at the same pace, so sharing a CU (one L1) turns
// most of their L2->L1 line fills into L1 hits.
//
// MF (SX == 1 only): the byte sums are taken by the matrix cores instead of the vector ALU.  Unpacking and adding a loaded
// dword costs the VALU 4 instructions = 16 SIMD clocks per 256 bytes, exactly what the L1 needs to deliver them: the
// VALU instance of the kernel is co-limited by the two.  v_mfma_i32_16x16x32_i8 computes D[i][j] += sum_k A[i][k] B[k][j]
// with B[8g .. 8g + 7][j] = the 8 bytes lane (j = lane & 15, g = lane >> 4) supplies: here the dwords of TWO beams loaded
// by that lane.  With the constant selector A[i][k] = ((k & 3) == (i & 3) && (k >> 3) == (i >> 2)) row 4g + b of D is, column
// by column, the sum of byte b of the two dwords of lane (j, g) -- and D[4g + b][j] is register b of lane (j, g): every lane
// gets the four byte sums of its own loads as 32-bit integers, one MFMA (16 SIMD clocks, its own pipe) per 512 loaded bytes,
// no unpacking and no 16-bit flushes.  (dwordx2 loads of two rows per lane would halve the load count, but the texture
// addresser handles them at half the byte rate: tools/tcp_ceiling.hip, 16.2 clocks per 512 bytes.)
template <int SX, int RY, int AW, bool MF>
__global__ __launch_bounds__(256 * AW) void k_score(const uint8_t * jobs, size_t stride, int n_jobs, int chunks,
                                                    int na_chunk, int tiles_max)
{
  // na_chunk = angle GROUPS (of AW angles) per unit
  const int per_unit = na_chunk * tiles_max;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int unit = (q / per_unit) * 8 + xcd;
  if (unit >= n_jobs * chunks) {return;}
  const int within = q % per_unit;
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)(unit / chunks) * stride);
  const int a_group = (unit % chunks) * na_chunk + within / tiles_max;
  const int tile = within % tiles_max;
  if (a_group * AW >= job.na || tile >= job.tiles_x * job.tiles_y) {return;}
  constexpr int PX = (SX == 1) ? kTileSpan : (kTileSpan + 1) / 2;   // poses per tile row
  constexpr int TY = 4 * RY;            // lattice rows per tile
  constexpr int NB = (SX == 1) ? 4 : 2; // byte positions per lane per row
  constexpr int UB = (RY >= 7) ? 4 : 8;   // beams per inner iteration
  const int tx = tile % job.tiles_x, ty = tile / job.tiles_x;
  const int x0 = tx * PX, y0 = ty * TY;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lx = lane & 15, ly = lane >> 4;
  const int sub = wave >> 2;                     // which of the AW angles this wave works on
  const int a = a_group * AW + sub;
  const bool live = a < job.na;                  // wave-uniform; dead waves only keep the barriers company
  const int tid = threadIdx.x & 255;             // thread index inside the angle's 4-wave team

  __shared__ int32_t s_tiles[AW][TY * PX];
  int32_t * s_tile = s_tiles[sub];
  for (int i = tid; i < TY * PX; i += 256) {s_tile[i] = 0;}
  __syncthreads();

  const int P = job.n_points;
  const int n_slow = live ? job.counts[kCountsPerAngle * a + kClasses] : 0;

  int32_t acc[RY][NB];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {acc[r][b] = 0;}
  }

  // this wave's alignment class and the byte of its first dword that belongs to pose x0
  const int cls = wave & 3;
  const int s = (cls + x0 * SX) & 3;
  // K2 keeps one list per (angle, class, scoring tile): the beams whose window has something in that tile
  const int lt = job.list_tiles;                 // tiles with lists of their own (1 = one list for all tiles)
  const size_t list_id = ((size_t)a * kClasses + cls) * lt + (lt > 1 ? tile : 0);
  // SX == 2: poses sit on every other byte, the even or the odd ones depending on s
  const uint32_t sel = (s & 1) ? 0x0c030c01u : 0x0c020c00u;
  uint32_t lo[RY], hi[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {lo[r] = 0; hi[r] = 0;}
  int since_flush = 0;
  auto flush = [&]() {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      if (SX == 1) {
        acc[r][0] += lo[r] & 0xffffu; acc[r][2] += lo[r] >> 16;
        acc[r][1] += hi[r] & 0xffffu; acc[r][3] += hi[r] >> 16;
      } else {
        acc[r][0] += lo[r] & 0xffffu; acc[r][1] += lo[r] >> 16;
      }
      lo[r] = 0; hi[r] = 0;
    }
    since_flush = 0;
  };
  // One list walk.  gbase = scalar part of the address (first byte of the lattice's window origin, moved back to the
  // dword boundary; >= -3: the allocations have zero bytes in front), row_bytes = bytes per lattice row step.
  // Offsets are fetched 64 at a time with one coalesced load and broadcast from the register with
  // v_readlane, so the inner loop is: 1 SALU add for the window address, RY saddr-form aligned
  // dword loads, 4 VALU per dword (2 at SX == 2).  Packed 16-bit partial sums are flushed every
  // 512 beams (512 * 100 < 65536).
  auto walk = [&](const gbyte * gbase, const gint * glist, const int n_list, const uint32_t row_bytes) {
    if (n_list <= 0) {return;}
    // per-lane byte offset of row r inside the window; rows beyond ny are clamped (sums discarded)
    uint32_t voff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      int yi = y0 + r * 4 + ly;
      yi = yi < job.ny ? yi : job.ny - 1;
      voff[r] = (uint32_t)(4 * lx) + (uint32_t)yi * row_bytes;
    }
    // a tile narrower than 61 poses (the second tile column of an 81-pose search holds 20): the lanes whose dword lies behind the
    // tile's last pose do not load -- their registers stay zero, their sums are never read
    const int np_tile = min(PX, job.nx - x0);
    const bool lane_on = 4 * lx <= s + (np_tile - 1) * SX;
    uint32_t w[UB][RY];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
#pragma unroll
      for (int r = 0; r < RY; ++r) {w[u][r] = 0u;}
    }
    for (int jc = 0; jc < n_list; jc += 64) {
      const int cnt = min(64, n_list - jc);
      const int32_t mine = (lane < cnt) ? glist[jc + lane] : 0;
      // UB beams per iteration: UB * RY loads are in flight before the first accumulate (the loop is
      // latency bound: ~1500 cycles per beam when every beam waits for its own loads)
      int k = 0;
      for (; k + UB <= cnt; k += UB) {
        if (lane_on) {
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const gbyte * wb = gbase + __builtin_amdgcn_readlane(mine, k + u);
#pragma unroll
            for (int r = 0; r < RY; ++r) {w[u][r] = *reinterpret_cast<const gu32 *>(wb + voff[r]);}
          }
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
        const gbyte * wbase = gbase + __builtin_amdgcn_readlane(mine, k);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
          const uint32_t w1 = lane_on ? *reinterpret_cast<const gu32 *>(wbase + voff[r]) : 0u;
          if (SX == 1) {
            lo[r] += w1 & 0x00ff00ffu;
            hi[r] += __builtin_amdgcn_perm(0u, w1, 0x0c030c01u);
          } else {
            lo[r] += __builtin_amdgcn_perm(0u, w1, sel);
          }
        }
      }
      since_flush += cnt;
      if (since_flush + 64 > 512) {flush();}
    }
  };
  // MF: 32-bit sums straight from the matrix cores; acc_mf[r][b] has the meaning of acc[r][b] (see the head of the kernel)
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i acc_mf[MF ? RY : 1];
#pragma unroll
  for (int t = 0; t < (MF ? RY : 1); ++t) {acc_mf[t] = v4i{0, 0, 0, 0};}
  auto walk_mf = [&](const gbyte * gbase, const gint * glist, const int n_list, const uint32_t row_bytes) {
    if (n_list <= 0) {return;}
    // selector A[i][k], lane (i = lane & 15, group = lane >> 4) holds k = 8 * group .. 8 * group + 7
    long sel;
    {
      const int i = lane & 15;
      sel = ((lane >> 4) == (i >> 2)) ? (long)(0x0000000100000001ull << (8 * (i & 3))) : 0l;
    }
    uint32_t voff[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) {
      int yi = y0 + r * 4 + ly;
      yi = yi < job.ny ? yi : job.ny - 1;
      voff[r] = (uint32_t)(4 * lx) + (uint32_t)yi * row_bytes;
    }
    for (int jc = 0; jc < n_list; jc += 64) {
      const int cnt = min(64, n_list - jc);
      const int32_t mine = (lane < cnt) ? glist[jc + lane] : 0;
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
        for (int u = 0; u < UB; u += 2) {
#pragma unroll
          for (int r = 0; r < RY; ++r) {
            const long pair = (long)(((unsigned long long)w[u + 1][r] << 32) | w[u][r]);
            acc_mf[r] = __builtin_amdgcn_mfma_i32_16x16x32_i8(sel, pair, acc_mf[r], 0, 0, 0);
          }
        }
      }
      for (; k < cnt; k += 2) {
        const bool two = k + 1 < cnt;
        const gbyte * wb0 = gbase + __builtin_amdgcn_readlane(mine, k);
        const gbyte * wb1 = two ? gbase + __builtin_amdgcn_readlane(mine, k + 1) : wb0;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
          const uint32_t w0 = *reinterpret_cast<const gu32 *>(wb0 + voff[r]);
          const uint32_t w1 = two ? *reinterpret_cast<const gu32 *>(wb1 + voff[r]) : 0u;
          const long pair = (long)(((unsigned long long)w1 << 32) | w0);
          acc_mf[r] = __builtin_amdgcn_mfma_i32_16x16x32_i8(sel, pair, acc_mf[r], 0, 0, 0);
        }
      }
    }
  };
  if (live) {
    // the beams whose window lies inside a grid row read the re-pitched copy K2 chose for them (every row segment in one
    // cache line), the few that wrap around the row end read the grid itself with its linear-index semantics
    if (MF) {
      if (job.grid2) {
        walk_mf(as_global(job.grid2) + (x0 * SX - s), as_global(job.fast2 + list_id * P), job.tcounts2[list_id],
          (uint32_t)job.pitch2 * (uint32_t)job.sy_cells);
      }
      walk_mf(as_global(job.grid) + ((int64_t)job.base0 + x0 * SX - s), as_global(job.fast + list_id * P), job.tcounts[list_id],
        (uint32_t)job.sy_ws);
    } else {
    if (job.grid2) {
      walk(as_global(job.grid2) + (x0 * SX - s), as_global(job.fast2 + list_id * P), job.tcounts2[list_id],
        (uint32_t)job.pitch2 * (uint32_t)job.sy_cells);
    }
    walk(as_global(job.grid) + ((int64_t)job.base0 + x0 * SX - s), as_global(job.fast + list_id * P), job.tcounts[list_id],
      (uint32_t)job.sy_ws);
    flush();
    }
  }

  // merge the four waves' partial sums: byte position j of the aligned tile row is pose (j - s) / SX
  // (MF: D[4 * group + b][column] of the MFMA is register b of lane (column, group) -- the lane that loaded the bytes)
  if (MF) {
#pragma unroll
    for (int r = 0; r < RY; ++r) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {acc[r][b] = acc_mf[r][b];}
    }
  }
#pragma unroll
  for (int r = 0; r < RY; ++r) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int j = 4 * lx + ((SX == 1) ? b : 2 * b + (s & 1));
      const int x = (j - s) / SX;
      if (j >= s && x < PX && acc[r][b] != 0) {atomicAdd(&s_tile[(r * 4 + ly) * PX + x], acc[r][b]);}
    }
  }

  if (n_slow > 0) {
    // per-pose range check exactly as GetResponse does it (Mapper.cpp:1192-1197); also the path of
    // non-linear lattices (bx/by are exact per-pose indices).  One pose per thread.
    const int32_t * slow = job.slow + (size_t)a * P;
    for (int p = tid; p < TY * PX; p += 256) {
      const int xi = x0 + p % PX, yi = y0 + p / PX;
      if (xi >= job.nx || yi >= job.ny) {continue;}
      const int64_t pose = (int64_t)job.bx[xi] + (int64_t)job.by[yi];
      int32_t sum = 0;
      for (int j = 0; j < n_slow; ++j) {
        const int64_t idx = pose + slow[j];
        if (idx >= 0 && idx < job.data_size) {sum += job.grid[idx];}
      }
      if (sum != 0) {atomicAdd(&s_tile[p], sum);}
    }
  }
  __syncthreads();

  // epilogue: sums, responses, best
  double best = 0.0;
  const size_t plane = (size_t)job.nx * job.ny;
  for (int p = tid; p < TY * PX; p += 256) {
    const int row = p / PX, col = p % PX;
    const int xi = x0 + col, yi = y0 + row;
    if (!live || xi >= job.nx || yi >= job.ny) {continue;}
    const int32_t sum = s_tile[p];
    const size_t o = (size_t)a * plane + (size_t)yi * job.nx + xi;
    job.sums[o] = sum;
    const double response = pose_response(job, sum, a, yi, xi);
    if (job.write_resp) {job.resp[o] = response;}
    best = response > best ? response : best;
  }
  // wave max -> workgroup (= tile of one angle) max -> one atomic per workgroup
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    const double o = __shfl_xor(best, sft);
    best = o > best ? o : best;
  }
  __shared__ double s_best[4 * AW];
  if (lane == 0) {s_best[wave] = best;}
  __syncthreads();
  if (live && (threadIdx.x & 255) == 0) {
    const double * wb = s_best + 4 * sub;
    double b = wb[0];
    b = wb[1] > b ? wb[1] : b; b = wb[2] > b ? wb[2] : b; b = wb[3] > b ? wb[3] : b;
    if (job.tile_best) {job.tile_best[(size_t)a * (job.tiles_x * job.tiles_y) + tile] = b;}   // K4 skips tiles without ties
    if (b > 0.0) {atomicMax(&job.out[0], (unsigned long long)__double_as_longlong(b));}
  }
}

void launch_score(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_tiles, int32_t max_na,
                  int32_t sx_variant, int32_t ry, void * stream, bool mfma)
{
  if (n_jobs <= 0 || max_tiles <= 0 || max_na <= 0) {return;}
  // AW = 4 (four adjacent angles sharing a CU) was measured: L1 hit rate 31 -> 38 %, no gain in time
  constexpr int AW = 1;
  const int groups = (max_na + AW - 1) / AW;      // angle groups per job
  // units of XCD-local work: whole jobs, or contiguous ranges of angle groups when jobs are scarce
  int chunks = 1;
  while (n_jobs * chunks * 2 <= 8 && chunks * 2 <= groups) {chunks *= 2;}
  const int na_chunk = (groups + chunks - 1) / chunks;
  const int units = n_jobs * chunks;
  const int units_per_xcd = (units + 7) / 8;
  const long long blocks = 8ll * units_per_xcd * na_chunk * max_tiles;
  dim3 grid((unsigned int)blocks);
  hipStream_t s = (hipStream_t)stream;
  // mfma: the matrix-core instance of the one-cell kernel (bit-identical sums; measured within +-5 % of the vector-ALU one --
  // 0.636 against 0.611 ms per 51 config-2 matches, 2.44 against 2.56 ms on the loop preset: the VALU is not what binds)
#define KH_SCORE(SXV, RYV, MFV) hipLaunchKernelGGL((k_score<SXV, RYV, AW, MFV>), grid, dim3(256 * AW), 0, s, d_jobs, stride, (int)n_jobs, chunks, na_chunk, (int)max_tiles)
  if (sx_variant == 2) {
    if (ry == 8) {KH_SCORE(2, 8, false);} else if (ry == 7) {KH_SCORE(2, 7, false);} else if (ry == 4) {KH_SCORE(2, 4, false);} else {KH_SCORE(2, 1, false);}
  } else if (!mfma) {
    if (ry == 8) {KH_SCORE(1, 8, false);} else if (ry == 7) {KH_SCORE(1, 7, false);} else if (ry == 4) {KH_SCORE(1, 4, false);} else {KH_SCORE(1, 1, false);}
  } else {
    if (ry == 8) {KH_SCORE(1, 8, true);} else if (ry == 7) {KH_SCORE(1, 7, true);} else if (ry == 4) {KH_SCORE(1, 4, true);} else {KH_SCORE(1, 1, true);}
  }
#undef KH_SCORE
}

// ---------------------------------------------------------------------------------------------
// K4: collect the poses whose response is within KT_TOLERANCE of the best (Mapper.cpp:807-808).
// The host sorts the (few) indices and does the libm averaging in the reference's order.
__global__ __launch_bounds__(256) void k_ties(const uint8_t * jobs, size_t stride)
{
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)blockIdx.y * stride);
  const size_t plane = (size_t)job.nx * job.ny;
  const double best = __longlong_as_double((long long)job.out[0]);
  uint32_t * tie_idx = reinterpret_cast<uint32_t *>(job.out + 2);
  if (job.coarse) {
    // search-space probabilities (Mapper.cpp:781-799): one plain store per cell instead of one atomic maximum per POSE in the
    // scoring kernel's epilogue (301 401 of them per config-2 match, all on the same 30 KB)
    (void)cell_maxima(job, (int)blockIdx.x, (int)gridDim.x, nullptr);
  }
  auto consider = [&](int a, int yi, int xi) {
    const size_t o = (size_t)a * plane + (size_t)yi * job.nx + xi;
    const double response = pose_response(job, job.sums[o], a, yi, xi);
    const double delta = response - best;
    const bool tie = delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06;
    if (tie) {
      const unsigned long long slot = atomicAdd(&job.out[1], 1ull);
      if (slot < (unsigned long long)kTieCap) {
        tie_idx[slot] = (uint32_t)(((size_t)yi * job.nx + xi) * job.na + a);
      }
    }
  };
  if (job.tile_best) {
    // only the scoring tiles whose own best response ties with the global one can hold a tie
    const int tiles = job.tiles_x * job.tiles_y;
    const int px = job.tile_px, ty_rows = 4 * job.ry;
    for (int pi = blockIdx.x; pi < job.na * tiles; pi += gridDim.x) {
      const double delta = job.tile_best[pi] - best;
      if (!(delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06)) {continue;}
      const int a = pi / tiles, tile = pi - a * tiles;
      const int x0 = (tile % job.tiles_x) * px, y0 = (tile / job.tiles_x) * ty_rows;
      for (int p = threadIdx.x; p < ty_rows * px; p += blockDim.x) {
        const int xi = x0 + p % px, yi = y0 + p / px;
        if (xi < job.nx && yi < job.ny) {consider(a, yi, xi);}
      }
    }
    return;
  }
  const size_t total = plane * job.na;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int a = (int)(o / plane);
    const int rem = (int)(o - (size_t)a * plane);
    const int yi = rem / job.nx, xi = rem - yi * job.nx;
    consider(a, yi, xi);
  }
}

// Fine passes: the host needs the raw sums of every angle at the best cell (ComputeAngularCovariance, Mapper.cpp:977-1025).
// Their volumes are tiny (3 x 3 x nA), so all of a batch's are packed into one buffer for ONE download instead of a
// copy per match (224 copies cost the stream 1.4 ms and the host 2.3 ms of enqueueing in the loop-closure batch).
__global__ __launch_bounds__(256) void k_gather_small(const uint8_t * jobs, size_t stride, int32_t * out, int small_stride)
{
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)blockIdx.x * stride);
  const int vol = job.nx * job.ny * job.na;
  if (job.coarse || vol > small_stride) {return;}
  int32_t * dst = out + (size_t)blockIdx.x * small_stride;
  for (int i = threadIdx.x; i < vol; i += blockDim.x) {dst[i] = job.sums[i];}
}

void launch_gather_small(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t * d_out, int32_t small_stride, void * stream)
{
  if (n_jobs <= 0 || small_stride <= 0) {return;}
  hipLaunchKernelGGL(k_gather_small, dim3(n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs, stride, d_out, (int)small_stride);
}

void launch_ties(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_poses, int32_t tile_pairs, void * stream)
{
  if (n_jobs <= 0 || max_poses <= 0) {return;}
  int blocks = (max_poses + 255) / 256;
  // every job of the launch has tile bests: pairs per job are few.  A batch: 16 workgroups per job, each taking four of the lattice's 64-cell
  // groups in turn -- the job's fields and the angle penalties fetched once per four groups (64 per job: k_ties 0.095 -> 0.081 ms per launch
  // of 51 config-2 matches, the scoring kernel beside it 0.435 -> 0.424)
  if (tile_pairs > 0) {blocks = std::min(tile_pairs, n_jobs >= 16 ? 16 : 64);}
  if (blocks > 1024) {blocks = 1024;}
  hipLaunchKernelGGL(k_ties, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs, stride);
}

}  // namespace kh

// =============================================================================================
// LDS-staged scoring path (windows of at most 61 bytes x 64 lattice rows: BASELINE config 2, the
// sequential preset, every fine search).
//
// The windowed kernel above is bound by the vector L1: every (angle, beam) window is 64 bytes x 64 rows = 4 KB of dword
// loads, the 32 KB L1 cannot keep the windows of the resident waves, and 64 B / clk / CU is all the L1 delivers however
// the lines are arranged.  But the windows of consecutive beams lie a few cells apart (median 3 - 7 cells at 5 mm), and
// those of the same beam at adjacent angles likewise: the UNION of the windows of some dozen consecutive beams at two
// adjacent angles is a rectangle of <= 192 bytes x <= 200 rows -- 300 - 1000 staged bytes per 4 KB window.  So:
//
//   K2' k_offsets_lds  one workgroup per (angle pair, job): bit-exact lookup table (as K2), empty-window test against the
//                      occupancy block map (as K2), then four builder waves (lane = beam) cut their quarter of the beams
//                      greedily into CHUNKS -- the longest run of consecutive beams whose windows' union fits one LDS
//                      region (wave-level prefix min / max of the window corners) -- and write per chunk a descriptor
//                      and, per angle, the LDS-relative window offsets sorted by alignment class.
//   K3' k_score_lds    512 threads = 2 angles x 4 row quarters, two workgroups per CU.  Per chunk: [barrier] LDS-DMA of the
//                      NEXT chunk's region into the other buffer (global_load_lds_dwordx4: no staging registers), then the
//                      current one is scored from LDS: aligned ds_read_b32 (pitch 192 B: the two rows a 32-lane half reads sit
//                      in disjoint banks), FOUR beams of one alignment class at a time, and the byte sums are taken by the
//                      matrix cores: v_mfma_i32_16x16x64_i8 with a constant selector adds, per lane, byte b of the four
//                      dwords it loaded into accumulator b -- one MFMA per 1 KB read, no unpacking, no VALU in the loop
//                      besides the four window addresses.  The alignment class (window start & 3) only decides which of a
//                      wave's four accumulator sets a beam adds to; the sets are merged with their shifts in the epilogue.
//
// LDS staging alone (round 1: ds_read_b32 + 4 VALU per dword) lost to the vector ALU, the matrix-core sums alone (round 3)
// to the L1; together neither is on the critical path.  What is: the waves' own instruction streams (a wave issues one instruction
// per turn of its SIMD, four waves per SIMD: round 5 took the step from 49 instructions to 35 and the kernel from 0.54 to 0.44 ms
// per 51 config-2 matches; reading 39 % fewer bytes -- the quad-skipping experiment, tools/patches/ -- made it SLOWER), and, for
// K2' and the tie kernel beside it, dependent memory round trips the compiler makes of conditional loads.
// =============================================================================================
namespace kh
{

struct ChunkDesc {int32_t beam_begin, g0, rows, windows, cnt[kChunkWords - 4];};      // cnt[q]: angle q's windows per alignment class, 4 x 8 bits
static_assert(sizeof(ChunkDesc) == kChunkWords * 4, "descriptor size");
static_assert(kGroupAngles >= 2 && kGroupAngles <= 4, "K3' runs kGroupAngles x 4 waves: at most 1024 threads");
static_assert(kLdsPitch % 16 == 0 && kLdsPitch % 128 == 64, "rows of a 32-lane half must fall into disjoint banks");
static_assert(kLdsRegionBytes % 1024 == 0 && kLdsRegionBytes >= kLdsRows * kLdsPitch, "one LDS-DMA instruction fills 1 KB");

constexpr int kNotFast = INT32_MIN;

// inclusive prefix minimum / maximum over the lanes of a wave, in registers: DPP row shifts inside the rows of 16 lanes, then the
// two row broadcasts (lane 15 of a row into the next row; lane 31 into the upper half).  A lane without a source keeps the
// operation's identity.  (Through __shfl_up every one of the 6 steps is an LDS round trip: the four scans of a builder step took
// ~2500 clocks.)
template <bool kMin>
__device__ __forceinline__ int wave_prefix(int v)
{
  constexpr int ident = kMin ? INT32_MAX : INT32_MIN;
  auto op = [](int x, int y) {return kMin ? (y < x ? y : x) : (y > x ? y : x);};
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));      // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));      // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));      // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));      // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1 and 3
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));      // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ int wave_prefix_min(int v, int) {return wave_prefix<true>(v);}
__device__ __forceinline__ int wave_prefix_max(int v, int) {return wave_prefix<false>(v);}

// K2'.  Thread t of the workgroup owns the beams t, t + 256, t + 512, ...: wave w therefore sees, round after round, a RUN of 64
// consecutive beams (beam = 256 * round + 64 * w + lane), which is all the chunk builder needs -- a chunk never spans more than 64
// beams.  Nothing but two counters lives in LDS (the lookup results stay in registers, the occupancy block map is read where it
// lies: 9 KB per job, shared by the 41 workgroups of the job in L2), so the workgroups of this kernel fit on compute units that
// already hold two workgroups of K3' (which leave 4 KB of LDS and a quarter of the registers free) and run in the issue slots
// K3' leaves idle, instead of taking whole compute units away from it.
__global__ __launch_bounds__(256) void k_offsets_lds(const uint8_t * jobs, size_t stride)
{
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)blockIdx.y * stride);
  const int group = blockIdx.x;
  const int a0 = group * kGroupAngles;
  if (a0 >= job.na) {return;}
  {
    // the job's result block starts from zero: every group's workgroup clears its share
    const int groups = (job.na + kGroupAngles - 1) / kGroupAngles;
    const int per = (job.out_words + groups - 1) / groups;
    const int hi = min(job.out_words, (group + 1) * per);
    for (int i = group * per + threadIdx.x; i < hi; i += blockDim.x) {job.out[i] = 0ull;}
  }
  const int P = job.n_points;
  __shared__ int32_t s_slow[kGroupAngles];
  if (threadIdx.x < kGroupAngles) {s_slow[threadIdx.x] = 0;}
  const uint32_t * const bmp = job.blockmap;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t bmin = job.base0;
  const int64_t bmax = (int64_t)job.base0 + (int64_t)(job.nx - 1) * job.sx + (int64_t)(job.ny - 1) * job.sy_ws;
  const int32_t xs = (job.nx - 1) * job.sx + 1, ys = (job.ny - 1) * job.sy_cells + 1;   // cells a window covers
  const int ws = job.ws, bm_w = job.bm_w, bm_h = job.bm_h, bshift = job.bshift, na = job.na, base0 = job.base0;
  const float inv_ws = 1.0f / (float)ws;
  const int64_t data_size = job.data_size, pad = job.pad;
  const double off_x = job.grid_off_x, off_y = job.grid_off_y, scale = job.scale;
  // rows K3' reads past a window's first row: its row waves cover 16, 32 or 64 lattice rows (lds_row_waves)
  const int row_waves = lds_row_waves(job.ny);
  const int read_rows = (16 * row_waves - 1) * job.sy_cells + 1;
  const int span_rows = ys;
  double cosine[kGroupAngles], sine[kGroupAngles];
#pragma unroll
  for (int q = 0; q < kGroupAngles; ++q) {
    cosine[q] = a0 + q < na ? job.cos_sin[2 * (a0 + q)] : 1.0;
    sine[q] = a0 + q < na ? job.cos_sin[2 * (a0 + q) + 1] : 0.0;
  }
  int32_t * const table = job.table;
  int32_t * const slow = job.slow;
  int32_t * const rel_out = job.rel;
  const int rounds = (P + 255) / 256;
  const int desc_cap = lds_desc_capacity(P);
  ChunkDesc * out = reinterpret_cast<ChunkDesc *>(job.chunks) + ((size_t)group * kLdsRanges + wave) * desc_cap;
  int n_out = 0;
  long long windows_total = 0;
  // this thread's beams: sensor-frame point and validity, all rounds in flight together
  constexpr int kMaxRounds = 8;                            // P <= 2048 on this path
  double plx[kMaxRounds], ply[kMaxRounds];
  bool pinv[kMaxRounds];
  // (unconditional loads at a clamped index, global address space: as `in ? load : default` every one of them sat in a branch of its
  // own with a wait behind it -- twenty-four dependent round trips in front of the first beam)
  {
    typedef __attribute__((address_space(1))) const uint8_t gflag;
    typedef __attribute__((address_space(1))) const double gdouble;
    gflag * const ginvalid = (gflag *)job.invalid;
    gdouble * const glocal = (gdouble *)job.local;
    uint8_t fl[kMaxRounds];
#pragma unroll
    for (int t = 0; t < kMaxRounds; ++t) {
      const int ic = min((int)threadIdx.x + 256 * t, P - 1);
      fl[t] = ginvalid[ic]; plx[t] = glocal[2 * ic]; ply[t] = glocal[2 * ic + 1];
    }
#pragma unroll
    for (int t = 0; t < kMaxRounds; ++t) {
      const int i = threadIdx.x + 256 * t;
      const bool in = t < rounds && i < P;
      pinv[t] = in ? fl[t] != 0 : true;
      plx[t] = in ? plx[t] : 0.0;
      ply[t] = in ? ply[t] : 0.0;
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxRounds; ++t) {
    if (t >= rounds) {break;}
    const int i = threadIdx.x + 256 * t;                  // this lane's beam of the round; the wave's run starts at i - lane
    const bool inr = i < P;
    int gx[kGroupAngles], gy[kGroupAngles];
#pragma unroll
    for (int q = 0; q < kGroupAngles; ++q) {
      gx[q] = 0; gy[q] = kNotFast;
      if (!inr || a0 + q >= na) {continue;}
      int32_t idx;
      if (pinv[t]) {
        idx = kInvalidScan;
      } else {
        const double lx = plx[t], ly = ply[t];
        // Karto.h:6879-6887: rotate, add the grid offset, WorldToGrid subtracts it again
        const double ox = cosine[q] * lx - sine[q] * ly;
        const double oy = sine[q] * lx + cosine[q] * ly;
        const double gxd = ((ox + off_x) - off_x) * scale;
        const double gyd = ((oy + off_y) - off_y) * scale;
        const int32_t gxi = d_to_int(d_round(gxd));
        const int32_t gyi = d_to_int(d_round(gyd));
        idx = (int32_t)((uint32_t)gxi + (uint32_t)gyi * (uint32_t)ws);   // base Grid::GridIndex, no ROI
        if (idx != kInvalidScan) {
          const bool off_grid = (int64_t)idx + bmax < 0 || (int64_t)idx + bmin >= data_size;
          // (a pose whose index falls off the array adds nothing in the reference, Mapper.cpp:1192-1197, and a zero here: the
          // allocation has job.pad zero bytes in front of and behind the grid)
          const bool inside = (int64_t)idx + bmin >= -pad && (int64_t)idx + bmax < data_size + pad;
          // the rectangle arithmetic needs the linear index to be exactly gx + gy * ws
          const bool exact = (int64_t)gxi + (int64_t)gyi * ws == (int64_t)idx && gxi > -(1 << 20) && gxi < (1 << 20);
          if (!off_grid) {
            if (inside && exact) {
              gx[q] = gxi; gy[q] = gyi;
              if (bmp) {
                // a window none of whose 32 x 32 blocks was touched by a stamp adds 0 to every pose of this angle: leave the
                // beam out (bit-identical sums).  Windows that wrap around the row end are kept (Appendix A.3).
                const int32_t start = (int32_t)((int64_t)idx + bmin);
                int32_t wy0 = (int32_t)((float)start * inv_ws);
                int32_t wx0 = start - wy0 * ws;
                while (wx0 < 0) {wx0 += ws; --wy0;}
                while (wx0 >= ws) {wx0 -= ws; ++wy0;}
                if (wx0 + xs <= ws && !window_has_blocks(bmp, bm_w, bm_h, wx0, wy0, wx0 + xs - 1, wy0 + ys - 1, bshift)) {
                  gy[q] = kNotFast;
                }
              }
            } else {
              (slow + (size_t)(a0 + q) * P)[atomicAdd(&s_slow[q], 1)] = idx;
            }
          }
        }
      }
      table[(size_t)(a0 + q) * P + i] = idx;
    }
    // ---- greedy chunks over this wave's run of 64 beams (lane = beam) ----
    const int run_lo = i - lane;                          // wave-uniform
    const int n_run = min(64, P - run_lo);                // beams of the run (<= 0: the run lies behind the scan)
    int begin = 0;
    while (begin < n_run) {
      const bool live_lane = lane >= begin && lane < n_run;
      int n = 0, x0 = 0, y0 = 0, rows = 0;
      bool any_chunk = false;
      for (;;) {
        int xmin = INT32_MAX, xmax = INT32_MIN, ymin = INT32_MAX, ymax = INT32_MIN;
#pragma unroll
        for (int q = 0; q < kGroupAngles; ++q) {
          if (live_lane && gy[q] != kNotFast) {
            xmin = min(xmin, gx[q]); xmax = max(xmax, gx[q]); ymin = min(ymin, gy[q]); ymax = max(ymax, gy[q]);
          }
        }
        xmin = wave_prefix_min(xmin, lane); xmax = wave_prefix_max(xmax, lane);
        ymin = wave_prefix_min(ymin, lane); ymax = wave_prefix_max(ymax, lane);
        const bool any = xmin != INT32_MAX;
        const int al = (int)(((int64_t)base0 + xmin) & 15);           // 16-byte alignment of the region's first row
        const int px0 = xmin - al;
        const bool fits = any && (xmax - px0 + kTileBytes <= kLdsPitch) && (ymax - ymin + read_rows <= kLdsRows);
        // the union only grows along the lanes: from `begin` on a run of ones, then zeros (lanes in front of `begin` count as ones)
        const unsigned long long okm = __ballot(!any || fits) | ((1ull << begin) - 1ull);
        const int first_bad = (okm == ~0ull) ? 64 : __ffsll((long long)~okm) - 1;
        n = first_bad - begin;
        if (n > 0) {
          // the chunk = lanes begin .. first_bad - 1; its rectangle is what the last of them sees
          x0 = __shfl(px0, first_bad - 1); y0 = __shfl(ymin, first_bad - 1);
          rows = __shfl(ymax, first_bad - 1) - y0 + span_rows;
          any_chunk = __shfl((int)any, first_bad - 1) != 0;
          break;
        }
        // the windows of beam `begin` at the group's angles are too far apart for one region: the last angle that still has one
        // goes the exact per-pose way (a window alone always fits; the first angle's follows only if it ever did not)
        if (lane == begin) {
          int qd = 0;
#pragma unroll
          for (int q = 1; q < kGroupAngles; ++q) {qd = gy[q] != kNotFast ? q : qd;}
#pragma unroll
          for (int q = 0; q < kGroupAngles; ++q) {
            if (q == qd) {
              (slow + (size_t)(a0 + q) * P)[atomicAdd(&s_slow[q], 1)] = gx[q] + gy[q] * ws;
              gy[q] = kNotFast;
            }
          }
        }
      }
      n = min(n, n_run - begin);
      if (any_chunk) {
        const bool in = lane >= begin && lane < begin + n;
        ChunkDesc d;
        d.beam_begin = run_lo + begin;
        d.g0 = y0 * ws + x0; d.rows = rows;
#pragma unroll
        for (int q = kGroupAngles; q < kChunkWords - 4; ++q) {d.cnt[q] = 0;}
        int windows = 0;
#pragma unroll
        for (int q = 0; q < kGroupAngles; ++q) {
          const int a = a0 + q;
          d.cnt[q] = 0;
          if (a >= na) {continue;}                             // wave-uniform
          const bool mine = in && gy[q] != kNotFast;
          const int rel = (gy[q] - y0) * kLdsPitch + (gx[q] - x0);
          // sort the chunk's offsets of this angle by alignment class: four contiguous segments
          const int cls = rel & 3;
          int offset = 0, my_rank = 0, packed = 0;
#pragma unroll
          for (int c = 0; c < kClasses; ++c) {
            const unsigned long long mask = __ballot(mine && cls == c);
            const int cnt = __popcll(mask);
            if (cls == c) {my_rank = offset + __popcll(mask & ((1ull << lane) - 1ull));}
            offset += cnt;
            packed |= cnt << (8 * c);
          }
          d.cnt[q] = packed;
          windows += offset;
          if (mine) {rel_out[(size_t)a * P + run_lo + begin + my_rank] = rel & ~3;}      // K3' reads aligned dwords; the class is the segment
        }
        d.windows = windows;
        windows_total += windows;
        if (lane == 0) {out[n_out] = d;}
        ++n_out;
      }
      begin += n;
    }
  }
  if (lane == 0) {
    job.chunk_counts[(size_t)group * kLdsRanges + wave] = n_out;
    // every window costs K3' sixteen wave-level ds_read_b32 (256 B each): the numerator of the roofline
    if (job.load_counter && windows_total) {atomicAdd(job.load_counter, (unsigned long long)(windows_total * 4 * row_waves));}
  }
  __syncthreads();
  if (threadIdx.x < kGroupAngles && a0 + (int)threadIdx.x < job.na) {
    job.counts[kCountsPerAngle * (a0 + threadIdx.x) + kClasses] = s_slow[threadIdx.x];
  }
}

void launch_offsets_lds(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, void * stream)
{
  if (n_jobs <= 0 || max_na <= 0) {return;}
  const int groups = (max_na + kGroupAngles - 1) / kGroupAngles;
  hipLaunchKernelGGL(k_offsets_lds, dim3(groups, n_jobs), dim3(256), 0, (hipStream_t)stream, d_jobs, stride);
}

// K3'.  S = grid cells per lattice step (x and y).  2 * NW waves: wave = q * NW + share: angle q of the pair, lattice rows
// 4 * RQ * share .. + 4 * RQ - 1 (RQ = 16 / NW); lane = (lx = lane & 15: one aligned dword of the 64-byte window row,
// ly = lane >> 4); rows yi = 4 * RQ * share + 4 * r + ly, r < RQ.  acc[c][r] = the four byte sums of the lane's dword over the
// class-c beams.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(1))) const void gvoid;
typedef int v4i __attribute__((ext_vector_type(4)));

#ifndef KH_PRIO
#define KH_PRIO 1
#endif
constexpr int kScoreLdsBlocksPerCu = (160 * 1024 - 4096) / (2 * kLdsRegionBytes);      // two regions per workgroup
static_assert(kScoreLdsBlocksPerCu >= 1 && kScoreLdsBlocksPerCu * kGroupAngles * 4 <= 16, "four waves per SIMD");
// (__launch_bounds__' second argument: the waves per SIMD the kernel must fit -- four: two workgroups of eight waves per compute unit)
template <int S, int NW, bool kFull>
__global__ __launch_bounds__(64 * kGroupAngles * NW, 4) void k_score_lds(const uint8_t * jobs, size_t stride, int n_jobs, int groups_max, int xcd_map)
{
  // kFull: every job of the launch has a lattice of more than 32 rows -- each of an angle's NW waves has its own rows and takes every
  // step (the bookkeeping of who takes which step leaves the chunk loop)
  constexpr int RQ = 16 / NW;                              // rows per lane (NW waves share the 64 rows of an angle's windows)
  constexpr int kThreadsPerAngle = 64 * NW;
  constexpr int PX = (S == 1) ? kTileSpan : (kTileSpan + 1) / 2;
  constexpr int kRowStep = 4 * S * kLdsPitch;              // bytes between two rows of a lane
  int job_index, group;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    job_index = (qb / groups_max) * 8 + xcd;               // XCD-aware: a job's angle groups share one L2
    group = qb % groups_max;
  } else {                                                 // fewer jobs than XCDs: spread the groups over all of them
    job_index = blockIdx.x / groups_max;
    group = blockIdx.x % groups_max;
  }
  if (job_index >= n_jobs) {return;}
  const CorrJob & job = *reinterpret_cast<const CorrJob *>(jobs + (size_t)job_index * stride);
  if (group * kGroupAngles >= job.na) {return;}
  extern __shared__ __attribute__((aligned(16))) uint32_t s_region[];     // two regions of kLdsRegionBytes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lx = lane & 15, ly = lane >> 4;
  const int q = wave / NW, quarter = wave % NW;
  // The NW waves of an angle: row_waves of them share the lattice rows (4 * RQ each), and when the lattice has fewer rows than
  // the NW waves cover together (31 rows: two waves) the waves left over take every other STEP of the same rows instead of
  // scoring rows nobody asked for: wave = (share of the rows, part of the steps).
  const int row_waves = kFull ? NW : lds_row_waves(job.ny) * (NW / 4);
  const int parts = kFull ? 1 : NW / row_waves;
  const int share = kFull ? quarter : quarter % row_waves, part = kFull ? 0 : quarter / row_waves;
  const int a = group * kGroupAngles + q;
  const bool live = a < job.na;
  const int P = job.n_points;
  const int range_len = lds_desc_capacity(P);         // descriptors one builder wave of K2' may write

  v4i acc[kClasses][RQ];
#pragma unroll
  for (int c = 0; c < kClasses; ++c) {
#pragma unroll
    for (int r = 0; r < RQ; ++r) {acc[c][r] = v4i{0, 0, 0, 0};}
  }
  // selector A[i][k] of the matrix product D = A B: lane (i = lane & 15, g = lane >> 4) holds k = 16 g .. 16 g + 15, the k
  // lane (j, g) of B supplies its four dwords for.  A[i][k] = (g == i >> 2 && (k & 3) == (i & 3)): row 4 g + b of D is, column
  // by column, the sum of byte b of the four dwords of lane (j, g) -- and D[4 g + b][j] is register b of lane (j, g).
  v4i sel;
  {
    const int i = lane & 15;
    const int one = ((lane >> 4) == (i >> 2)) ? (int)(0x01010101u & (0xffu << (8 * (i & 3)))) : 0;
    sel = v4i{one, one, one, one};
  }
  // byte offset of this lane's first row inside a window (the other rows are immediates: fixed pitch)
  const int lanebase = 4 * lx + (4 * RQ * share + ly) * S * kLdsPitch;

  const gbyte * gwin = as_global(job.grid) + job.base0;
  const gint * grel = as_global(job.rel + (size_t)(live ? a : 0) * P);
  // descriptors and counts are wave-uniform: read through the scalar cache (constant address space: K2' wrote them in an
  // earlier launch), so the chunk loop's control flow stays on the scalar unit
  typedef const __attribute__((address_space(4))) int32_t cint;
  cint * cdescs = (cint *)(job.chunks + (size_t)group * kLdsRanges * range_len * kChunkWords);
  cint * ccounts = (cint *)(job.chunk_counts + (size_t)group * kLdsRanges);
  struct Chunk {int beam_begin, g0, rows, packed;};

  // LDS-DMA of one region: a wave instruction moves 64 units of 16 bytes (unit u = row * (pitch / 16) + column block) to 1 KB
  // of LDS starting at a wave-uniform address.  (job.ws is read once, in front of the loop: a load of it inside would put an
  // s_waitcnt vmcnt(0) -- i.e. a wait for the PREVIOUS DMA -- in front of every DMA instruction.)
  constexpr int kUnitsPerRow = kLdsPitch / 16;
  constexpr int kWaves = kGroupAngles * NW;
  constexpr int kDmaPerWave = (kLdsRegionBytes / 1024 + kWaves - 1) / kWaves;
  const int ws = job.ws;
  // the grid offset (relative to the region's first byte) of the unit this lane moves in the wave's t-th DMA instruction is the same
  // for every chunk: computed once (a division per instruction and chunk otherwise); past the region's last unit it is clamped to
  // that unit's (offsets grow with the unit number): the tail block re-reads the last unit and stays in bounds
  uint32_t dma_off[kDmaPerWave];
#pragma unroll
  for (int t = 0; t < kDmaPerWave; ++t) {
    const int u = 64 * (wave + kWaves * t) + lane;
    const int row = u / kUnitsPerRow, col = u - row * kUnitsPerRow;
    dma_off[t] = (uint32_t)(row * ws + 16 * col);
  }
  auto issue_dma = [&](const Chunk & d, int buf) {
    const gbyte * src = gwin + d.g0;
    const int nblk = (d.rows * kUnitsPerRow + 63) >> 6;
    const uint32_t o_last = (uint32_t)((d.rows - 1) * ws + 16 * (kUnitsPerRow - 1));
    lds_u32 * dst = (lds_u32 *)s_region + buf * (kLdsRegionBytes / 4) + wave * 256;
#pragma unroll
    for (int t = 0; t < kDmaPerWave; ++t) {
      if (wave + kWaves * t >= nblk) {break;}
      const uint32_t o = dma_off[t] < o_last ? dma_off[t] : o_last;
      __builtin_amdgcn_global_load_lds((gvoid *)(src + o), (__attribute__((address_space(3))) void *)(dst + kWaves * t * 256), 16, 0, 0);
    }
  };
  // this wave's offsets of a chunk: angle q's class-sorted run, lane k = k-th window
  auto load_rel = [&](const Chunk & d) -> int32_t {
    const int cnt = (d.packed & 0xff) + ((d.packed >> 8) & 0xff) + ((d.packed >> 16) & 0xff) + ((d.packed >> 24) & 0xff);
    return (live && lane < cnt) ? grel[d.beam_begin + lane] : 0;
  };
  // One STEP = four windows of the class-sorted run: 4 * RQ single ds_read_b32 (RQ rows x four beams), then RQ MFMAs into the
  // accumulator set of their class.  The reads are written as asm so that every result lands in its slot of an MFMA operand
  // tuple (left to itself the compiler pairs the rows of a beam into ds_read2st64_b32 and then shuffles the halves into the
  // tuples with a dozen v_mov per step); the wait is explicit, the loads' results are tied to it.
  // A wave issues one instruction per turn of its SIMD whatever the instruction is, so the step is kept to the instructions it
  // needs -- four readlanes, four adds, sixteen reads, a wait, four MFMAs, three of loop control -- and free of branches.  The
  // last step of a class may hold fewer than four beams: its missing slots read the step's first window again and the selector
  // of THAT step's MFMAs leaves their dwords out, so it is the same code with a different selector (round 4: a zero strip in LDS
  // and a branch per missing slot, in every step: 49 instructions and four taken branches where 32 and one do).
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)((char *)s_region);
  const uint32_t lds_lane = lds_base + (uint32_t)lanebase;
#define KH_DSR(dst, addr, r) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"((r) * kRowStep) : "memory")
#define KH_STEP(selector) do { \
    int32_t w[RQ][4]; \
    _Pragma("unroll") for (int r = 0; r < RQ; ++r) {KH_DSR(w[r][0], a0, r); KH_DSR(w[r][1], a1, r); KH_DSR(w[r][2], a2, r); KH_DSR(w[r][3], a3, r);} \
    if (RQ == 4) { \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]), "+v"(w[1][3]), \
        "+v"(w[RQ - 2][0]), "+v"(w[RQ - 2][1]), "+v"(w[RQ - 2][2]), "+v"(w[RQ - 2][3]), "+v"(w[RQ - 1][0]), "+v"(w[RQ - 1][1]), "+v"(w[RQ - 1][2]), "+v"(w[RQ - 1][3])); \
    } else { \
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]), "+v"(w[1][3])); \
    } \
    _Pragma("unroll") for (int r = 0; r < RQ; ++r) { \
      acc[c][r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(selector, (v4i{w[r][0], w[r][1], w[r][2], w[r][3]}), acc[c][r], 0, 0, 0); \
    } \
  } while (0)
  int step_base = 0;
  auto score = [&](const Chunk & d, int buf, int32_t rels) {
    if (!live) {return;}
    const uint32_t base = lds_lane + (uint32_t)(buf * kLdsRegionBytes);
    int off = 0;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
      const int cnt = (d.packed >> (8 * c)) & 0xff;
      // this wave's steps of the class: those whose running number (over the classes and chunks of the angle) is `part`
      // modulo `parts`
      const int first = kFull ? 0 : (part - step_base) & (parts - 1);
      if (!kFull) {step_base += (cnt + 3) >> 2;}
      int k = 4 * first;
      for (; k + 4 <= cnt; k += 4 * parts) {
        const int i0 = off + k;
        const uint32_t a0 = base + (uint32_t)__builtin_amdgcn_readlane(rels, i0);
        const uint32_t a1 = base + (uint32_t)__builtin_amdgcn_readlane(rels, i0 + 1);
        const uint32_t a2 = base + (uint32_t)__builtin_amdgcn_readlane(rels, i0 + 2);
        const uint32_t a3 = base + (uint32_t)__builtin_amdgcn_readlane(rels, i0 + 3);
        KH_STEP(sel);
      }
      if (k < cnt) {
        const int i0 = off + k, rem = cnt - k;             // 1 .. 3 windows
        const uint32_t a0 = base + (uint32_t)__builtin_amdgcn_readlane(rels, i0);
        const uint32_t a1 = base + (uint32_t)__builtin_amdgcn_readlane(rels, rem > 1 ? i0 + 1 : i0);
        const uint32_t a2 = base + (uint32_t)__builtin_amdgcn_readlane(rels, rem > 2 ? i0 + 2 : i0);
        const uint32_t a3 = a0;
        const v4i sel_tail = v4i{sel[0], rem > 1 ? sel[1] : 0, rem > 2 ? sel[2] : 0, 0};
        KH_STEP(sel_tail);
      }
      off += cnt;
    }
  };
#undef KH_STEP
#undef KH_DSR

  // walk the chunks of the four beam ranges in order: [barrier] DMA of the next region, score this one; the descriptor after
  // the next is already on its way through the scalar cache
  static_assert(kLdsRanges == 4, "the four chunk counts are held in scalar registers");
  // (chunk number -> builder range and place in it without branches: the ranges' ends)
  const int end0 = ccounts[0], end1 = end0 + ccounts[1], end2 = end1 + ccounts[2], n_chunks = end2 + ccounts[3];
  int fetched = 0;
  auto next_chunk = [&](Chunk & d) -> bool {
    if (fetched >= n_chunks) {return false;}
    const int range = (fetched >= end0 ? 1 : 0) + (fetched >= end1 ? 1 : 0) + (fetched >= end2 ? 1 : 0);
    const int start = fetched >= end2 ? end2 : fetched >= end1 ? end1 : fetched >= end0 ? end0 : 0;
    cint * w = cdescs + ((size_t)range * range_len + (fetched - start)) * kChunkWords;
    d.beam_begin = w[0]; d.g0 = w[1]; d.rows = w[2]; d.packed = w[4 + q];
    ++fetched;
    return true;
  };
  Chunk cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0}, aft = {0, 0, 0, 0};
  bool have = next_chunk(cur);
  bool more = have && next_chunk(nxt);
  int buf = 0;
  int32_t rel_cur = 0, rel_nxt = 0;
  if (have) {issue_dma(cur, 0); rel_cur = load_rel(cur);}
  while (have) {
    __syncthreads();                                 // this region landed (the barrier drains the DMA); the other one is free
    // (the region's DMA pieces issued one in front of every alignment class instead of together here: 0.455 against 0.444 ms)
    if (more) {issue_dma(nxt, buf ^ 1); rel_nxt = load_rel(nxt);}
    const bool after = more && next_chunk(aft);
    // the scoring steps above the waves of K2' / K4 that share the compute unit (the other staging set's launches): their issue slots
    // are the scoring kernel's own time, the side kernels have a whole scoring launch to finish in (0.441 -> 0.427 ms per launch, same box)
    __builtin_amdgcn_s_setprio(KH_PRIO);
    score(cur, buf, rel_cur);
    __builtin_amdgcn_s_setprio(0);
    cur = nxt; nxt = aft; rel_cur = rel_nxt; buf ^= 1; have = more; more = after;
  }
  __syncthreads();

  const size_t plane = (size_t)job.nx * job.ny;
  const int ta = tid % kThreadsPerAngle;             // thread index inside the angle's team of NW waves
  // merge the four accumulator sets of every wave with the classes' shifts, the four waves of an angle in LDS, then one pose per thread
  int32_t * s_tile = reinterpret_cast<int32_t *>(s_region) + q * (64 * PX);
  if (S == 1 && kFull) {
    // In registers: pose x = 4 lx + b of a row collects byte x + c of class c -- register (b + c) & 3 of this lane, or of the next
    // lane of the row of 16 (DPP row_shl:1) where b + c reaches the next dword.  Every pose of the wave's rows comes out of exactly
    // one lane: plain stores, no zeroing pass in front (through LDS atomics the merge was 64 of them per lane: 14 of a wave's 200
    // thousand clocks).
    if (live) {
#pragma unroll
      for (int r = 0; r < RQ; ++r) {
        int32_t * const row = s_tile + (4 * RQ * share + 4 * r + ly) * PX + 4 * lx;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int32_t total = acc[0][r][b];
#pragma unroll
          for (int c = 1; c < kClasses; ++c) {
            const int32_t v = acc[c][r][(b + c) & 3];
            total += (b + c) < 4 ? v : __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true);      // row_shl:1: lane i reads lane i + 1
          }
          if (4 * lx + b < PX) {row[b] = total;}
        }
      }
    }
  } else {
    for (int i = ta; i < 64 * PX; i += kThreadsPerAngle) {s_tile[i] = 0;}
    __syncthreads();
    if (live) {
#pragma unroll
      for (int c = 0; c < kClasses; ++c) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int j = 4 * lx + b;                              // byte position inside the aligned window row
            const int x = (j - c) / S;
            const bool pose = j >= c && ((j - c) % S) == 0 && x < PX;
            if (pose && acc[c][r][b] != 0) {atomicAdd(&s_tile[(4 * RQ * share + 4 * r + ly) * PX + x], acc[c][r][b]);}
          }
        }
      }
    }
  }
  // (the pose loop's inputs are fetched here, the accumulators' registers being free: on their way across the barrier)
  constexpr int kPosesPerThread = (64 * PX + kThreadsPerAngle - 1) / kThreadsPerAngle;
  // everything the pose loop reads from the job, once: behind a store the compiler has to assume the job block itself changed
  // and re-reads every field through the scalar cache, a wait per field and pose
  const int nx = job.nx, ny = job.ny;
  const bool penal = job.do_penalize != 0, wr_resp = job.write_resp != 0;
  const double denom = job.denom;
  int32_t * const sums_out = job.sums + (size_t)(live ? a : 0) * plane;
  double * const resp_out = wr_resp ? job.resp + (size_t)(live ? a : 0) * plane : nullptr;
  const int32_t * const jbx = job.bx, * const jby = job.by;
  const uint8_t * const jgrid = job.grid;
  const int64_t data_size = job.data_size;
  // the distance penalties of this thread's poses first, all in flight together
  double dpen[kPosesPerThread];
  const double apen = (live && penal) ? job.ang_pen[a] : 1.0;
#pragma unroll
  for (int k = 0; k < kPosesPerThread; ++k) {
    const int p = ta + kThreadsPerAngle * k;
    const int yi = p / PX, xi = p % PX;
    dpen[k] = (live && penal && xi < nx && yi < ny) ? job.dist_pen[yi * nx + xi] : 1.0;
  }
  __syncthreads();
  const int n_slow = live ? job.counts[kCountsPerAngle * a + kClasses] : 0;
  const int32_t * slow = job.slow + (size_t)(live ? a : 0) * P;
  double best = 0.0;
#pragma unroll
  for (int k = 0; k < kPosesPerThread; ++k) {
    const int p = ta + kThreadsPerAngle * k;
    const int yi = p / PX, xi = p % PX;
    if (!live || xi >= nx || yi >= ny) {continue;}
    int32_t sum = s_tile[p];
    if (n_slow > 0) {
      // per-pose range check exactly as GetResponse does it (Mapper.cpp:1192-1197)
      const int64_t pose = (int64_t)jbx[xi] + (int64_t)jby[yi];
      for (int j = 0; j < n_slow; ++j) {
        const int64_t idx = pose + slow[j];
        if (idx >= 0 && idx < data_size) {sum += jgrid[idx];}
      }
    }
    const size_t o = (size_t)yi * nx + xi;
    sums_out[o] = sum;
    // pose_response with the penalties at hand (same operations: Mapper.cpp:1204, 671-685)
    double response = (double)sum / denom;
    if (penal) {
      const double delta = response - 0.0;
      const bool is_zero = delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06;
      if (!is_zero) {response *= (dpen[k] * apen);}
    }
    if (wr_resp) {resp_out[o] = response;}
    best = response > best ? response : best;
  }
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    const double o = __shfl_xor(best, sft);
    best = o > best ? o : best;
  }
  // angle maximum -> K4 skips the angles without ties (one scoring tile per angle on this path); one atomic per angle
  __shared__ double s_best[kGroupAngles * NW];
  if (lane == 0) {s_best[wave] = best;}
  __syncthreads();
  if (live && ta == 0) {
    const double * wb = s_best + NW * q;
    double b = wb[0];
#pragma unroll
    for (int t = 1; t < NW; ++t) {b = wb[t] > b ? wb[t] : b;}
    if (job.tile_best) {job.tile_best[a] = b;}
    if (b > 0.0) {atomicMax(&job.out[0], (unsigned long long)__double_as_longlong(b));}
  }
}

void launch_score_lds(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, int32_t sx_variant, bool full_rows, void * stream)
{
  if (n_jobs <= 0 || max_na <= 0) {return;}
  const int groups = (max_na + kGroupAngles - 1) / kGroupAngles;
  const int xcd_map = n_jobs >= 8 ? 1 : 0;
  const int jobs_per_xcd = (n_jobs + 7) / 8;
  const long long blocks = xcd_map ? 8ll * jobs_per_xcd * groups : (long long)n_jobs * groups;
  constexpr int kDyn = 2 * kLdsRegionBytes;                                  // two regions
  // four waves per angle (16 rows each, 512 threads, 4 waves per SIMD); eight (8 rows each) saturate the SIMD's VALU port: DESIGN.md
  // (per device: a group's members on other devices launch this kernel from their own threads)
  static std::atomic<unsigned long long> attr_done[4] = {{0}, {0}, {0}, {0}};
  allow_dynamic_lds(reinterpret_cast<const void *>(k_score_lds<1, 4, true>), kDyn, attr_done[0]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_score_lds<2, 4, true>), kDyn, attr_done[1]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_score_lds<1, 4, false>), kDyn, attr_done[2]);
  allow_dynamic_lds(reinterpret_cast<const void *>(k_score_lds<2, 4, false>), kDyn, attr_done[3]);
  hipStream_t s = (hipStream_t)stream;
#define KH_SCORE_LDS(SV, FV) hipLaunchKernelGGL((k_score_lds<SV, 4, FV>), dim3((unsigned int)blocks), dim3(64 * kGroupAngles * 4), kDyn, s, d_jobs, stride, (int)n_jobs, groups, xcd_map)
  if (sx_variant == 2) {
    if (full_rows) {KH_SCORE_LDS(2, true);} else {KH_SCORE_LDS(2, false);}
  } else {
    if (full_rows) {KH_SCORE_LDS(1, true);} else {KH_SCORE_LDS(1, false);}
  }
#undef KH_SCORE_LDS
}

}  // namespace kh
