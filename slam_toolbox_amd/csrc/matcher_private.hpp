// Private definitions of the scan matcher's host side, shared by matcher_host.cpp (batches) and matcher_seq.cpp (the fused
// path of ONE MatchScan).  Not part of the ABI (include/karto_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/karto_hip.h"
#include "kh_internal.hpp"
#include "host_pool.hpp"

namespace kh
{

extern thread_local std::string g_last_error;
void set_error(const std::string & s);

#define KH_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error(std::string(#call) + ": " + hipGetErrorString(e_));                          \
      return KH_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

// ---- exact scalar helpers (Math.h) ----------------------------------------------------------
// cos and sin of ONE angle, the way the reference's Release build computes them: GCC (-O1 and up) merges a cos(a) / sin(a)
// pair into one sincos(a) call, and glibc's sincos is NOT bit-identical to its cos and sin everywhere (a = 0.11462314399891493:
// cos(a) = 0.9934379567501339, sincos(a) gives 0.993437956750134).  Every place where the reference takes both of the same
// angle goes through here, so that the library does not depend on whether ITS compiler merges the pair (clang does not).
inline void ref_sincos(double a, double * s, double * c) {::sincos(a, s, c);}
constexpr double kTolerance = 1e-06;                 // Math.h:41
constexpr double kPi = 3.14159265358979323846;       // Math.h:31
constexpr double k2Pi = 6.28318530717958647692;      // Math.h:32
constexpr double kPi180 = 0.01745329251994329577;    // Math.h:34
constexpr double kMaxVariance = 500.0;               // Mapper.cpp:52
constexpr double kDistanceGain = 0.2;                // Mapper.cpp:53
constexpr double kAngleGain = 0.2;                   // Mapper.cpp:54

inline double round_half_away(double v) {return v >= 0.0 ? std::floor(v + 0.5) : std::ceil(v - 0.5);}
inline int32_t to_int32(double v)
{
  if (!(v > -2147483649.0 && v < 2147483648.0)) {return INT32_MIN;}
  return static_cast<int32_t>(v);
}
inline bool double_equal(double a, double b)
{
  const double delta = a - b;
  return delta < 0.0 ? delta >= -kTolerance : delta <= kTolerance;
}
inline double normalize_angle(double angle)   // Math.h:181-202
{
  while (angle < -kPi) {
    if (angle < -k2Pi) {angle += static_cast<uint32_t>(angle / -k2Pi) * k2Pi;} else {angle += k2Pi;}
  }
  while (angle > kPi) {
    if (angle > k2Pi) {angle -= static_cast<uint32_t>(angle / k2Pi) * k2Pi;} else {angle -= k2Pi;}
  }
  return angle;
}
inline double normalize_angle_difference(double minuend, double subtrahend)   // Math.h:213-224
{
  while (minuend - subtrahend < -kPi) {minuend += k2Pi;}
  while (minuend - subtrahend > kPi) {minuend -= k2Pi;}
  return minuend;
}
struct Cell {int32_t x, y;};
inline Cell world_to_grid(double scale, double ox, double oy, double wx, double wy)   // Karto.h:4421-4436
{
  const double gx = (wx - ox) * scale;
  const double gy = (wy - oy) * scale;
  return Cell{to_int32(round_half_away(gx)), to_int32(round_half_away(gy))};
}
inline size_t align_up(size_t v, size_t a) {return (v + a - 1) / a * a;}

// ---- per-correlate host context (what finalisation needs) -----------------------------------
struct CorrHost
{
  int32_t slot = 0;
  int32_t P = 0, nx = 0, ny = 0, na = 0;
  double center[3] = {0, 0, 0};
  double off_x = 0, off_y = 0, res_x = 0, res_y = 0, ang_off = 0, ang_res = 0;
  bool fine = false, penalize = false;
  std::vector<double> x_poses, y_poses, angles, dist_pen, ang_pen;
  int32_t lt_alloc = 1;          // tile lists the slot's `fast` buffer was sized for
  std::vector<int32_t> bx, by;
  double denom = 1.0;
};

struct StageLayout {size_t bx, by, dist_pen, ang_pen, cos_sin, local, invalid, total;};

// Staging and bookkeeping of one in-flight sub-batch of CorrelateScan jobs (a handle owns two: pipelining)
struct CorrBatch
{
  std::vector<CorrHost> ctx;
  std::vector<StageLayout> lay;
  size_t stride = 0, out_words = 0;
  int32_t tile_pairs = 0;
  int32_t max_na = 0, max_tiles = 0, max_poses = 0, sx_variant = -1, ry = -1;
  bool uniform_kernel = true, use_lds = false;
  // staging (pinned host + device mirror) for the jobs; pinned result mirror; small fine-pass volumes
  uint8_t * h_stage = nullptr; uint8_t * d_stage = nullptr; size_t cap_stage = 0, cap_dstage = 0;
  unsigned long long * h_out = nullptr; size_t cap_hout = 0;   // words
  unsigned long long * d_out = nullptr; size_t cap_dout = 0;   // words: one contiguous result block per job
  int32_t * h_sums = nullptr; size_t cap_hsums = 0;          // fine passes: packed small volumes (pinned) ...
  int32_t * d_small = nullptr; size_t cap_dsmall = 0; size_t small_stride = 0;   // ... and their device staging
  hipEvent_t ev[2] = {nullptr, nullptr};      // around the scoring kernel (profiling)
  hipEvent_t evs[4] = {nullptr, nullptr, nullptr, nullptr};   // around the table / list kernel (K2) and the tie kernel (K4) (profiling)
  hipEvent_t done = nullptr;                  // everything of the sub-batch, downloads included
  hipEvent_t up = nullptr, kdone = nullptr;   // chunked batches: tables + lists ready (side stream) / scoring finished (main stream)
  hipStream_t side = nullptr;                 // side stream of this staging set (uploads, K2, K4, downloads of its chunks)
};

struct Slot
{
  uint8_t * d_grid = nullptr;        // first grid byte (256-byte aligned); the allocation has kGridPad zero bytes either side
  uint8_t * d_grid_alloc = nullptr;
  uint32_t * d_blockmap = nullptr;   // bm_w x bm_h occupancy blocks of this grid (cleared and marked with it)
  uint8_t * d_grid2 = nullptr;       // re-pitched copies A and B (CorrJob::grid2), allocated at the first search that profits
  uint8_t * d_grid2_alloc = nullptr;
  int32_t copy_kind = 0;             // 0 none, 1 copies A / B of the grid, 2 column-decimated copies (RasterJob::copy_kind)
  int32_t * d_prev_work = nullptr;   // tiles the previous rasterisation touched (what has to be zeroed in the copies)
  double off_x = 0.0, off_y = 0.0;      // CoordinateConverter offset of this slot's grid
  // correlate scratch
  int32_t * d_table = nullptr, * d_fast = nullptr, * d_slow = nullptr, * d_counts = nullptr;
  int32_t * d_tcounts = nullptr; size_t cap_tcounts = 0, cap_fast = 0;
  size_t cap_table = 0, cap_counts = 0;
  int32_t * d_chunks = nullptr, * d_chunk_counts = nullptr; size_t cap_chunks = 0, cap_chunk_counts = 0;
  int32_t * d_sums = nullptr; double * d_resp = nullptr; size_t cap_volume = 0, cap_resp = 0;
  // raster scratch: per-point stamp flags (K0 FindValidPoints + the order-dependent rule), cell table of the order-dependent rule
  uint8_t * d_ractive = nullptr; size_t cap_ractive = 0;
  uint32_t * d_hkeys = nullptr; int32_t * d_hvals = nullptr; uint8_t * d_hstate = nullptr; int32_t * d_hnbr = nullptr;
  size_t cap_hkeys = 0, cap_hvals = 0, cap_hstate = 0, cap_hnbr = 0;
  double * d_tile_best = nullptr; size_t cap_tile_best = 0;
  int32_t * d_rtiles = nullptr;      // tile_count | tile_cursor | n_work(+pad) | tile_start | work   (first three zeroed per raster)
  int32_t * d_rlists = nullptr; size_t cap_rlists = 0;   // cell_xy (2 np) | list (4 np) | rank (4 np)
  // first-point rasteriser (matcher_seq.hip): the table over the region of interest (kept clean between rasterisations), the
  // candidate records, the control words, the (tile, start, count) records of kseq_tile
  int32_t * d_first = nullptr; size_t cap_first = 0; bool first_clean = false;
  int32_t * d_cand = nullptr; size_t cap_cand = 0;
  int32_t * d_seqctl = nullptr;
  int32_t * d_work2 = nullptr; size_t cap_work2 = 0;
  // last correlate (for the introspection calls)
  CorrHost last;
  bool has_last = false;
  bool volume_stale = false;         // the stored volume was overwritten by an off-lattice re-score (introspection reports it)
  // what ComputePositionalCovariance reads: the search-space probabilities of the last COARSE search (Mapper.cpp:726-732, 781-799)
  CorrHost last_coarse; std::vector<double> last_lattice; bool has_last_coarse = false;
};

}  // namespace kh

namespace kh {struct SeqState;}
using namespace kh;

struct kh_matcher
{
  double search_size = 0, resolution = 0, smear = 0, range_threshold = 0;
  int32_t width = 0, height = 0, ws = 0, data_size = 0;
  int32_t roi_x = 0, roi_y = 0, roi_w = 0, roi_h = 0, kernel_size = 0, side = 0;
  double scale = 0;
  std::vector<uint8_t> kernel;
  std::vector<Cell> footprint100;       // kernel cells equal to 100 (relative offsets)
  kh_match_params params;
  int32_t device = 0, max_batch = 1;
  hipStream_t stream = nullptr;
  uint8_t * d_kernel = nullptr;
  std::vector<Slot> slots;
  CorrBatch batch[2];
  // raster staging: the distinct base scans' unfiltered points (pinned mirror + device arena), the jobs' scan lists and
  // the (job, scan) work items of K0 (one int32 block), the jobs
  double * h_arena = nullptr; double * d_arena = nullptr; size_t cap_harena = 0, cap_darena = 0;
  int32_t * h_meta = nullptr; int32_t * d_meta = nullptr; size_t cap_hmeta = 0, cap_dmeta = 0;
  RasterJob * h_rjobs = nullptr; RasterJob * d_rjobs = nullptr;
  bool keep_responses = false;
  bool force_chunks = std::getenv("KH_FORCE_CHUNKS") != nullptr;       // kh_matcher_set_debug bit 3: chunk every batch of >= 128 (tests)
  bool dense_score = false;        // kh_matcher_set_debug bit 2: do not skip beams whose window is empty
  int32_t bm_w = 0, bm_h = 0, bshift = kBlockShift;    // occupancy block map: words per row, rows, log2 of the block side
  int32_t rt_w = 0, rt_h = 0;      // rasteriser tiles over the grid
  int32_t pitch2 = 0, copy_b = 0;  // dual-copy layout: row pitch (multiple of 128) and byte offset of copy B
  int32_t pad_rows = 0;            // zero rows in front of and behind every slot's grid and copies (CorrJob::pad)
  size_t grid_pad = 0;             // the same in bytes of the grid's own pitch, rounded up to 256, plus kGridPad
  int32_t pitch_d = 0, copy_q = 0; // column-decimated copies: row pitch and bytes of one of the four; copy_q 0 = too large for int32 offsets
  bool dual_copy = true;           // kh_matcher_set_debug bit 4 switches the re-pitched copies off (measurements)
  bool mfma_score = std::getenv("KH_K3_MFMA") != nullptr;   // kh_matcher_set_debug bit 5: byte sums on the matrix cores (k_score<.., MF>)
  bool lds_score = false;          // kh_matcher_set_debug bit 1: LDS-staged scoring path for every search it can take (default: the large ones)
  bool windowed_score = false;     // kh_matcher_set_debug bit 6: never (the windowed kernel k_score scores everything)
  uint8_t * d_tab = nullptr;       // padded image of the smear kernel for kseq_tile (kernels of >= 8 x 8 cells)
  bool table_raster = false;       // batches are rasterised by the first-point kernels too (the slots' tables fit: kh_matcher_create)
  bool no_seq = false;             // kh_matcher_set_debug bit 7: one MatchScan takes the general (batch) path instead of the fused one
  kh::SeqState * seq = nullptr;    // state of the fused path of ONE MatchScan (matcher_seq.cpp), made at its first use
  // profiling
  bool profiling = false;
  double score_ms = 0, raster_ms = 0; int64_t score_launches = 0, raster_launches = 0, score_jobs = 0;
  double offsets_ms = 0, ties_ms = 0;      // the kernels either side of the scoring kernel, same launches (kh_matcher_profile_side)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long * d_load_counter = nullptr;    // see CorrJob::load_counter (only handed to the jobs while profiling)

  double grid_resolution() const {return 1.0 / scale;}   // Karto.h:4518-4521
};

namespace kh
{
template <class T>
int ensure_device(T *& p, size_t & cap, size_t need, hipStream_t stream)
{
  if (need <= cap) {return KH_OK;}
  if (p) {
    KH_HIP(hipStreamSynchronize(stream));
    KH_HIP(hipFree(p));
    p = nullptr;
  }
  size_t n = std::max(need, cap + cap / 2);
  KH_HIP(hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)));
  cap = n;
  return KH_OK;
}
template <class T>
int ensure_pinned(T *& p, size_t & cap, size_t need, hipStream_t stream)
{
  if (need <= cap) {return KH_OK;}
  if (p) {
    KH_HIP(hipStreamSynchronize(stream));
    KH_HIP(hipHostFree(p));
    p = nullptr;
  }
  size_t n = std::max(need, cap + cap / 2);
  KH_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(T), hipHostMallocDefault));
  cap = n;
  return KH_OK;
}


// ---- requests ------------------------------------------------------------------------------------
struct RasterReq {int32_t slot; const kh_scan * query; const kh_scan * base; int32_t n_base;};
struct CorrReq
{
  int32_t slot;
  const kh_scan * scan;
  double center[3];
  double off_x, off_y, res_x, res_y, ang_off, ang_res;
  bool penalize, fine;
  // results
  double mean[3]; double cov[9]; double response; int status;
};
struct WalkGeometry {double center[3], off_x, off_y, res_x, res_y, ang_res;};

// internal result of the fused sequential path: "take the general path for this step" (never leaves the library)
constexpr int kNeedGeneric = 1000;

// shape of one prepared job: what picks the scoring kernel instance of a launch
struct JobShape {int32_t sx = 1, ry = 1, tiles = 1, lds = 0;};
// where a finished job's results lie (host memory)
struct ResultView
{
  const unsigned long long * out = nullptr;   // result block (kOutHeaderWords header + lattice maxima), see kh_internal.hpp
  const int32_t * small = nullptr;            // fine passes: the packed volume [a][y][x] (nullptr: not downloaded)
  CorrJob * h_job = nullptr;                  // staged job (host mirror) and its device copy: the off-lattice re-score edits them
  uint8_t * d_job = nullptr;
  size_t stride = 0;
  bool device_work = true;                    // false: a finalisation that would have to touch the stream returns kNeedGeneric instead
};

StageLayout stage_layout(int32_t P, int32_t nx, int32_t ny, int32_t na, bool penalize);
int pick_ry(int32_t ny);
int init_ctx(const CorrReq & q, CorrHost & c);
// allow_copies = false: the slot is not given re-pitched copies of its grid by this call (the fused path of one match scores from
// the grid itself; copies the slot already has are kept in step all the same)
int ensure_slot_scratch(kh_matcher * m, const CorrReq & q, CorrHost & c, bool allow_copies = true);
void prepare_job(kh_matcher * m, const CorrReq & q, CorrHost & c, const StageLayout & L, uint8_t * hb, uint8_t * db,
  unsigned long long * d_out, size_t out_words, size_t n_launch, bool lds_always, bool lds_never, JobShape & shape);
int finalize_job(kh_matcher * m, CorrReq & q, CorrHost & c, const ResultView & v);
int raster_batch(kh_matcher * m, const std::vector<RasterReq> & reqs);
int correlate_batch(kh_matcher * m, std::vector<CorrReq> & reqs);
// matcher_seq.cpp: ONE MatchScan through the fused kernels.  *coarse_done / *fine_done say which passes it finished
// A caller inside the library (the mapper's Process) may leave the QUERY scan's readings unfinished when it calls kh_matcher_match
// and hand over the function that finishes them: the fused path runs it behind the rasteriser's launches -- which need the query's
// sensor pose and nothing else of it -- so that LocalizedRangeScan::Update of the new scan (1081 sincos: 15 us) runs while the GPU
// works; every other path runs it before it reads the query.  Per thread; consumed by the next kh_matcher_match_batch on the thread.
struct QueryHook
{
  std::function<void()> fn;
  void run() {if (fn) {std::function<void()> f; f.swap(fn); f();}}
};
QueryHook & pending_query_hook();
int seq_match(kh_matcher * m, const kh_scan * query, const kh_scan * base, int32_t n_base, bool penalize, bool refine,
  double mean[3], double cov[9], double * response, int * status, bool * coarse_done, bool * fine_done);
void seq_destroy(kh_matcher * m);
// counters of the fused path since the handle was made (kh_matcher_seq_stats)
constexpr int kSeqStatWords = 8;
enum {kSeqStatCalls = 0, kSeqStatFineOnDevice = 1, kSeqStatFineFallback = 2, kSeqStatFineMismatch = 3, kSeqStatCoarseFallback = 4, kSeqStatFusedScore = 5};
const int64_t * seq_stats(const kh_matcher * m);
void fill_raster_job(const kh_matcher * m, const Slot & s, const double * pose, int32_t n_points, size_t npad, RasterJob & j);
// the slot's tables of the first-point rasteriser for a job of n_points (allocated / grown / cleaned as needed) and the handle's
// tile-kernel image; fills them into the job
int ensure_seq_tables(kh_matcher * m, Slot & s, int32_t n_points, RasterJob & j);

}  // namespace kh
