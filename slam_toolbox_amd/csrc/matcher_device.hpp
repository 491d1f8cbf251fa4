// Device helpers shared by the scan matcher's kernel files (matcher_kernels.hip: batches; matcher_seq.hip: the fused path of ONE
// MatchScan).  Build with -ffp-contract=off: the roundings must see the reference's IEEE operations, unfused.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include "kh_internal.hpp"
#include "lds_attr.hpp"

namespace kh
{

// ---------------------------------------------------------------------------------------------
// exact helpers (mirror Math.h:87-90 and the x86-64 double->int32 conversion)
__device__ __forceinline__ double d_round(double v) {return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5);}
__device__ __forceinline__ int32_t d_to_int(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) {return INT32_MIN;}   // cvttsd2si "integer indefinite"
  return (int32_t)v;
}
// job point p -> its world coordinates: base scan by bisection of the prefix (<= a few dozen scans), then the arena
__device__ __forceinline__ double2 job_point(const RasterJob & job, int p)
{
  if (job.uniform_n > 0) {                      // the usual case: one laser, every scan has the same number of beams
    const int k = p / job.uniform_n;
    return reinterpret_cast<const double2 *>(job.scan_ptr[k])[p - k * job.uniform_n];
  }
  int lo = 0, hi = job.n_scans;                 // scan_prefix[lo] <= p < scan_prefix[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (job.scan_prefix[mid] <= p) {lo = mid;} else {hi = mid;}
  }
  return reinterpret_cast<const double2 *>(job.scan_ptr[lo])[p - job.scan_prefix[lo]];
}
// CoordinateConverter::WorldToGrid (Karto.h:4421-4436) + the ROI test of AddScan (Mapper.cpp:1083-1088)
__device__ __forceinline__ bool roi_cell(const RasterJob & job, double2 w, int32_t & gx, int32_t & gy)
{
  const double gxd = (w.x - job.off_x) * job.scale;
  const double gyd = (w.y - job.off_y) * job.scale;
  gx = d_to_int(d_round(gxd)); gy = d_to_int(d_round(gyd));
  return (gx >= 0 && gx < job.roi_w) && (gy >= 0 && gy < job.roi_h);
}
// ---------------------------------------------------------------------------------------------
// response of one pose from its integer sum: GetResponse's normalisation (Mapper.cpp:1204) and the
// odometry penalty (Mapper.cpp:671-685).  Shared by K3 and K4 so both see identical bits.
__device__ __forceinline__ double pose_response(const CorrJob & job, int32_t sum, int a, int yi, int xi)
{
  double response = (double)sum / job.denom;
  if (job.do_penalize) {
    const double delta = response - 0.0;
    const bool is_zero = delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06;   // math::DoubleEqual, Math.h:135-139
    if (!is_zero) {
      response *= (job.dist_pen[yi * job.nx + xi] * job.ang_pen[a]);
    }
  }
  return response;
}

// global-address-space views: keeps the hot loads on global_load (vmcnt only) instead of flat_load
typedef __attribute__((address_space(1))) uint8_t gbyte;
typedef __attribute__((address_space(1))) int32_t gint;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ const gbyte * as_global(const uint8_t * p) {return (const gbyte *)p;}
__device__ __forceinline__ const gint * as_global(const int32_t * p) {return (const gint *)p;}

// Does any stamp footprint overlap the window [x_lo, x_hi] x [y_lo, y_hi] (grid cells)?  One bit per block of 2^bshift cells squared, rows padded
// by a word; rows above and below the array hold nothing.  The probes of up to nine block rows -- a 61-cell window on 8 x 8 blocks -- are
// independent loads issued together: as a loop of load, wait, or (what the compiler makes of the plain form) the test of one window
// was nine dependent trips to the L2, and K2' -- two windows per beam, five beams per thread -- 0.19 ms per 64 matches.
__device__ __forceinline__ bool window_has_blocks(const uint32_t * bmp, int bm_w, int bm_h, int x_lo, int y_lo, int x_hi, int y_hi, int bshift)
{
  const int bx0 = x_lo >> bshift, bx1 = x_hi >> bshift;
  const int by0 = max(y_lo, 0) >> bshift, by1 = min(y_hi >> bshift, bm_h - 1);
  const int wi = bx0 >> 5, sh = bx0 & 31, nb = bx1 - bx0 + 1;
  if (nb > 32) {return true;}
  const unsigned long long span = (1ull << nb) - 1ull;
  typedef __attribute__((address_space(1))) const uint32_t gword;
  gword * const words = (gword *)bmp + wi;
  constexpr int kRowsAtOnce = 9;
  unsigned long long any = 0;
  for (int by = by0; by <= by1; by += kRowsAtOnce) {
    uint32_t lo[kRowsAtOnce], hi[kRowsAtOnce];
#pragma unroll
    for (int k = 0; k < kRowsAtOnce; ++k) {
      gword * row = words + (size_t)min(by + k, by1) * bm_w;         // past the last row: the last row again
      lo[k] = row[0]; hi[k] = row[1];
    }
#pragma unroll
    for (int k = 0; k < kRowsAtOnce; ++k) {any |= ((unsigned long long)hi[k] << 32) | lo[k];}
  }
  return ((any >> sh) & span) != 0;
}

// FindValidPoints, data parallel inside one scan (workgroup per scan).  The state machine hops from trigger to trigger -- a
// trigger is the first reading more than 0.1 m from the current anchor, and it becomes the next anchor -- so its path is a
// walk along next(i) = "first reading after i more than 0.1 m from reading i", which every lane can evaluate for its own
// readings.  Which readings the walk visits (reachability from the first valid reading) comes from pointer doubling in
// LDS: 11 rounds for <= 2048 readings instead of one dependent hop per trigger (several hundred per scan when the beams
// are long: 156 us for the 20 running scans of a sequential match with the hop-by-hop kernel k_find_valid).  The
// side-of-line sign of every visited trigger and the fate of every run follow in parallel: reading i is emitted iff the
// first trigger after it lies on the viewpoint's side (Mapper.cpp:1145-1160); the tail after the last trigger never is.
// Same comparisons, same operand order as the sequential form: bit-identical flags.
// (device function: k_find_valid_par runs it per (job, scan) item, the fused sequential path's kseq_prep per scan of its one job and
// then goes on with the scan's points -- they stay in LDS as P[i], their flags as flags_lds[i] = reach[i].)
// s_fv: max_n double2 + 3 * (max_n + 64) int32 + 2 * (max_n + 64) bytes of dynamic LDS.  Any block size that is a multiple of 64.
template <bool kWaveSearch>
__device__ __forceinline__ void find_valid_scan(const double2 * pts, const int n, uint8_t * out, const double vx, const double vy,
  const int max_n, double2 * s_fv, uint8_t *& flags_lds, long long * dbg = nullptr)
{
  // (dbg: wall_clock64 stamps of the phases, one workgroup's thread 0 -- measurements)
#define KH_FV_STAMP(k) do {if (dbg && threadIdx.x == 0) {dbg[k] = (long long)wall_clock64();}} while (0)
  // one workgroup per scan: a lane owns every 256th reading, so the divergent forward scans of next() cost a lane four or
  // five readings' worth of its slowest neighbour instead of seventeen (one wave per scan: 56 us for 20 scans)
  const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
  const int stride_i = max_n + 64;                        // ints per pointer array
  double2 * P = s_fv;
  int32_t * nxt0 = reinterpret_cast<int32_t *>(P + max_n);
  int32_t * nxa = nxt0 + stride_i;
  int32_t * nxb = nxa + stride_i;
  uint8_t * reach = reinterpret_cast<uint8_t *>(nxb + stride_i);
  uint8_t * keep = reach + stride_i;
  flags_lds = reach;
  __shared__ int s_pos0;
  __shared__ unsigned long long s_mask[40];               // triggers of every chunk of 64 readings (max_n <= 2048 -> 32 chunks)
  __shared__ int s_later[40];                             // first trigger in the chunks behind chunk c, -1 = none
  if (tid == 0) {s_pos0 = n;}
  KH_FV_STAMP(0);
  for (int i = tid; i < n; i += nthreads) {P[i] = pts[i]; reach[i] = 0; keep[i] = 0;}
  __syncthreads();
  KH_FV_STAMP(1);
  const double min_square_distance = 0.1 * 0.1;          // math::Square(0.1), folded in double like the host does
  // the first reading without a NaN coordinate is the first anchor (Mapper.cpp:1127-1136)
  {
    int mine = n;
    for (int i = tid; i < n; i += nthreads) {
      if (!isnan(P[i].x) && !isnan(P[i].y)) {mine = i; break;}
    }
    if (mine < n) {atomicMin(&s_pos0, mine);}
  }
  // next(i): the trigger that follows if reading i is the anchor (n = none)
  if (kWaveSearch) {
    // a WAVE per anchor, the lanes on the 64 readings behind it (ballot, first set bit): one step for almost every anchor.  With a
    // lane per anchor the kernel waits for its slowest lane -- a reading with dozens of neighbours within 0.1 m (an obstacle half
    // a metre away, a run of NaN) walks them one LDS round trip at a time, and the other lanes' work does not shorten that walk.
    // 64 x the comparisons: for one job on an idle chip, not for batches.
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    for (int i = wave; i < n; i += nwaves) {
      const double fx = P[i].x, fy = P[i].y;
      int j = n;
      if (!(isnan(fx) || isnan(fy))) {                       // a NaN reading is never an anchor (nothing is "farther" than NaN)
        for (int base = i + 1; base < n; base += 64) {
          const int jj = base + lane;
          bool hit = false;
          if (jj < n) {
            const double dx = fx - P[jj].x, dy = fy - P[jj].y;
            hit = dx * dx + dy * dy > min_square_distance;
          }
          const unsigned long long mask = __ballot(hit);
          if (mask) {j = base + __builtin_ctzll(mask); break;}
        }
      }
      if (lane == 0) {nxt0[i] = j;}
    }
  } else {
    for (int i = tid; i < n; i += nthreads) {
      const double fx = P[i].x, fy = P[i].y;
      int j = (isnan(fx) || isnan(fy)) ? n : i + 1;         // a NaN reading is never an anchor (nothing is "farther" than NaN)
      for (; j < n; ++j) {
        const double dx = fx - P[j].x, dy = fy - P[j].y;
        if (dx * dx + dy * dy > min_square_distance) {break;}
      }
      nxt0[i] = j;
    }
  }
  if (tid == 0) {nxt0[n] = n; reach[n] = 0;}
  __syncthreads();
  KH_FV_STAMP(2);
  const int pos0 = s_pos0;
  if (pos0 >= n) {
    for (int i = tid; i < n; i += nthreads) {out[i] = 0; flags_lds[i] = 0;}
    __syncthreads();
    return;
  }
  if (tid == 0) {reach[pos0] = 1;}
  __syncthreads();
  // reachability from pos0 by pointer doubling
  const int32_t * cur = nxt0;
  int32_t * nxt_w = nxa;
  for (int span = 1; span < n; span <<= 1) {
    for (int i = tid; i <= n; i += nthreads) {
      const int j = cur[i];
      if (i < n && reach[i] && j < n) {reach[j] = 1;}
      nxt_w[i] = j < n ? cur[j] : n;
    }
    __syncthreads();
    cur = nxt_w;
    nxt_w = (nxt_w == nxa) ? nxb : nxa;
  }
  KH_FV_STAMP(3);
  // every visited trigger: which side of the line viewpoint -> anchor it lies on (its anchor is the visited reading whose
  // next() it is)
  for (int i = tid; i < n; i += nthreads) {
    const int j = nxt0[i];
    if (reach[i] && j < n) {
      const double fx = P[i].x, fy = P[i].y, cx = P[j].x, cy = P[j].y;
      const double a = vy - fy;
      const double b = fx - vx;
      const double cc = fy * vx - fx * vy;
      const double ss = cx * a + cy * b + cc;
      keep[j] = ss < 0.0 ? 0 : 1;
    }
  }
  // reading i belongs to the run that ends at the first trigger after it
  const int n_chunks = (n + 63) >> 6;
  for (int base = 64 * (tid >> 6); base < n; base += nthreads) {     // wave w takes chunks w, w + 4, ...
    const int i = base + lane;
    const bool trig = i < n && i != pos0 && reach[min(i, n - 1)];
    const unsigned long long mask = __ballot(trig);
    if (lane == 0) {s_mask[base >> 6] = mask;}
  }
  __syncthreads();
  KH_FV_STAMP(4);
  if (tid == 0) {
    int later = -1;
    for (int c = n_chunks - 1; c >= 0; --c) {
      s_later[c] = later;
      if (s_mask[c]) {later = 64 * c + __builtin_ctzll(s_mask[c]);}
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += nthreads) {
    const int c = i >> 6, l = i & 63;
    const unsigned long long above = l < 63 ? (s_mask[c] >> (l + 1)) : 0ull;
    const int j = above ? i + 1 + __builtin_ctzll(above) : s_later[c];
    const uint8_t flag = j >= 0 ? keep[j] : (uint8_t)0;
    out[i] = flag;
    flags_lds[i] = flag;
  }
  __syncthreads();
  KH_FV_STAMP(5);
#undef KH_FV_STAMP
}

// The search-space probabilities (Mapper.cpp:781-799): the best response over the angles of every cell of the lattice, from the
// stored sums with pose_response -- for the 64-cell groups first_group, first_group + group_stride, ... of the job; every thread
// of a 256-thread block must call it.  Written to the job's result block (and to `mirror`, when given: the same lattice in
// host-coherent memory).  Returns the largest response the calling thread saw among its cells (complete in the threads of wave 0).
// The exact response costs a double-precision division per angle.  The maximum over the angles is found in two passes: a
// single-precision key sum x angle penalty first (the distance penalty is common to the cell; the key orders the PENALISED
// responses up to a relative 3e-7), then the exact response of the angles whose key lies within 1e-5 of the largest -- one or
// two of the 81 -- and of every angle whose sum is so small that its response may be at or below 1e-6: those skip the penalty
// (math::DoubleEqual(response, 0), Mapper.cpp:671-685), do not follow the key, and can exceed a penalised response when the
// minimum penalties are small (they are user parameters; a key floor keeps angles with a zero penalty in the running).
// Work split: 64 cells at a time (lane = cell: coalesced rows of the sums volume), the four waves a quarter of the angles
// each -- a thread's sums are read ONCE, all in flight together, and kept in registers for both passes.
__device__ __forceinline__ double cell_maxima(const CorrJob & job, const int first_group, const int group_stride, unsigned long long * mirror)
{
  const size_t plane = (size_t)job.nx * job.ny;
  const int na = job.na, nxp = job.nx;
  const int32_t * const sums = job.sums;
  unsigned long long * const probs = job.out + kOutHeaderWords;
  const bool penal = job.do_penalize != 0;
  const double denom = job.denom;
  const int32_t v_small = (int32_t)(1.0e-6 * denom) + 1;      // sums up to here may give a response at or below 1e-6
  constexpr int kSlice = 32;                             // angles per wave held in registers
  __shared__ float s_key[4][64];
  __shared__ double s_max[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int per = (na + 3) / 4;
  double seen = 0.0;
  if (per <= kSlice) {
    const int a_lo = slice * per;
    // the wave's angle penalties: ONE load (lane t = angle a_lo + t), handed out by readlane where they are used.  (Read where
    // they are used -- job.ang_pen[a_lo + t] inside the two loops below -- every one of them was a flat load with its own wait:
    // two dozen dependent memory round trips per workgroup, 0.14 ms per launch of 64 matches for a kernel that moves 77 MB.)
    const double lane_pen = (penal && lane < per && a_lo + lane < na) ? job.ang_pen[a_lo + lane] : 1.0;
    auto angle_penalty = [&](int t) -> double {
      const long long bits = __double_as_longlong(lane_pen);
      const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), t), hi = __builtin_amdgcn_readlane((int)(bits >> 32), t);
      return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    const gint * const gsums = as_global(sums);
    for (int cell0 = first_group * 64; cell0 < (int)plane; cell0 += group_stride * 64) {
      const int cell = cell0 + lane;
      const bool valid = cell < (int)plane;
      int32_t v[kSlice];
#pragma unroll
      for (int t = 0; t < kSlice; ++t) {v[t] = (valid && t < per && a_lo + t < na) ? gsums[(size_t)(a_lo + t) * plane + cell] : 0;}
      // (the cell's distance penalty with the same batch of loads)
      const double dpen = (valid && penal) ? job.dist_pen[cell] : 1.0;
      float kmax = 0.0f;
#pragma unroll
      for (int t = 0; t < kSlice; ++t) {
        if (t < per && a_lo + t < na) {
          const float key = fmaxf((float)v[t] * (float)angle_penalty(t), v[t] > 0 ? 1e-30f : 0.0f);
          kmax = key > kmax ? key : kmax;
        }
      }
      s_key[slice][lane] = kmax;
      __syncthreads();
      kmax = fmaxf(fmaxf(s_key[0][lane], s_key[1][lane]), fmaxf(s_key[2][lane], s_key[3][lane]));
      double m = 0.0;
      if (valid && kmax > 0.0f) {
        const float thresh = kmax * (1.0f - 1e-5f);
#pragma unroll
        for (int t = 0; t < kSlice; ++t) {
          if (t < per && a_lo + t < na && v[t] > 0) {
            const double ap = angle_penalty(t);
            const float key = fmaxf((float)v[t] * (float)ap, 1e-30f);
            if (key >= thresh || v[t] <= v_small) {
              // pose_response with the penalties at hand (same operations: Mapper.cpp:1204, 671-685)
              double response = (double)v[t] / denom;
              if (penal) {
                const double delta = response - 0.0;
                const bool is_zero = delta < 0.0 ? delta >= -1e-06 : delta <= 1e-06;
                if (!is_zero) {response *= (dpen * ap);}
              }
              m = response > m ? response : m;
            }
          }
        }
      }
      s_max[slice][lane] = m;
      __syncthreads();
      if (slice == 0 && valid) {
        const double m01 = s_max[0][lane] > s_max[1][lane] ? s_max[0][lane] : s_max[1][lane];
        const double m23 = s_max[2][lane] > s_max[3][lane] ? s_max[2][lane] : s_max[3][lane];
        const double mm = m01 > m23 ? m01 : m23;
        probs[cell] = (unsigned long long)__double_as_longlong(mm);
        if (mirror) {mirror[cell] = (unsigned long long)__double_as_longlong(mm);}
        seen = mm > seen ? mm : seen;
      }
      __syncthreads();
    }
  } else {
    // more than 128 angles: a thread per cell walks them all, eight loads at a time
    for (int cell = first_group * 256 + (int)threadIdx.x; cell < (int)plane; cell += group_stride * 256) {
      const int yi = cell / nxp, xi = cell - yi * nxp;
      float kmax = 0.0f;
      for (int a0 = 0; a0 < na; a0 += 8) {
        int32_t v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {v[t] = a0 + t < na ? sums[(size_t)(a0 + t) * plane + cell] : 0;}
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          if (a0 + t < na) {
            const float key = fmaxf((float)v[t] * (penal ? (float)job.ang_pen[a0 + t] : 1.0f), v[t] > 0 ? 1e-30f : 0.0f);
            kmax = key > kmax ? key : kmax;
          }
        }
      }
      double m = 0.0;
      if (kmax > 0.0f) {
        const float thresh = kmax * (1.0f - 1e-5f);
        for (int a = 0; a < na; ++a) {
          const int32_t v = sums[(size_t)a * plane + cell];
          const float key = fmaxf((float)v * (penal ? (float)job.ang_pen[a] : 1.0f), 1e-30f);
          if (v > 0 && (key >= thresh || v <= v_small)) {
            const double response = pose_response(job, v, a, yi, xi);
            m = response > m ? response : m;
          }
        }
      }
      probs[cell] = (unsigned long long)__double_as_longlong(m);
      if (mirror) {mirror[cell] = (unsigned long long)__double_as_longlong(m);}
      seen = m > seen ? m : seen;
    }
  }
  return seen;
}

}  // namespace kh
