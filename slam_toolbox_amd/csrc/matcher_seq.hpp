// The fused path of ONE MatchScan (ScanMatcher::MatchScan, Mapper.cpp:534-639 -- the only way the reference's API is ever
// called: Mapper.cpp:2714-2717, 1472, 1653).  Kernel argument blocks shared by matcher_seq.cpp (host) and matcher_seq.hip.
//
// The general path (matcher_host.cpp) is built for batches: per stage one launch over all jobs, descriptors uploaded, results
// downloaded, the host between the coarse and the fine pass.  One job through it is 19 launches, three copies, two stream
// waits and 0.41 ms, every kernel a chain of dependent memory round trips on a nearly empty chip.  Here the same arithmetic
// (bit for bit: tests/test_seq_gpu.py) is laid out for one job:
//
//   kseq_prep    FindValidPoints per scan (1024 threads per scan), the kept readings' grid cells, the FIRST point of every cell
//                (atomicMin into a per-slot table over the region of interest: no hash table, no probing), Grid::Clear of
//                the tiles the previous match wrote -- descriptors arrive as kernel arguments, nothing is uploaded
//   kseq_links   per first point the first points of the cells in its 100-footprint (four independent loads)
//   kseq_bin     ONE workgroup: the order-dependent "cell already 100" rule as a fixpoint over state bytes in LDS, then tile
//                counts, list starts and lists with LDS counters only (the batch path: five launches, global atomics)
//   kseq_tile    the stamps, tile by tile (eight waves per tile, two dependent loads in front of the first stamp; smear kernels
//                below 8 x 8 cells: k_raster_tile of the batch path), k_repitch* for slots with copies; its last workgroups pull
//                the host's exact tables (computed while the kernels above run) into device memory and zero volume and result
//   kseq_score   lookup table + scoring in one launch (K2 + K3), the beams of an angle cut into slices of 64 (17 x as many
//                workgroups as K3 has for one job), sums added to the volume
//   kseq_cells   per-cell maxima (the search-space probabilities) and the best response
//   kseq_ties    ONE workgroup: ties; when the coarse pass has exactly one best pose, the fine search's centre (from host-made
//                tables indexed by the coarse angle: no libm on the device) and lattice
//   kseq_fine    the fine pass around it, a wave per (angle, 64 beams)
//   kseq_done    ONE workgroup: the fine pass's responses, best and ties; results into host-coherent memory, then a flag the
//                host polls -- no host round trip between coarse and fine, no copies
#pragma once
#include <cstdint>
#include "kh_internal.hpp"

namespace kh
{

constexpr int32_t kSeqMaxScans = 128;      // base scans of one match (pointers travel as kernel arguments: 1.5 KB of the 4 KB)
constexpr int32_t kSeqMaxReadings = 2048;  // readings per scan (k_find_valid_par's LDS working set)
constexpr int32_t kSeqMaxPoints = 131072;  // job points: one state byte each in LDS (the host checks the sum against the LDS)
constexpr int32_t kSeqMaxTiles = 16384;    // rasteriser tiles: one counter each in LDS
constexpr int32_t kSeqMaxFine = 1024;      // poses of the fine volume
constexpr int32_t kSeqCandWords = 8;       // int32 per stamp candidate: point, cx, cy, selected | four neighbour points
constexpr int32_t kFirstNone = INT32_MAX;
constexpr int32_t kSeqSlice = 64;          // beams per workgroup of kseq_score
constexpr int32_t kSeqCtlWords = 16;

struct SeqPrepArgs
{
  RasterJob job;                           // (scan_ptr / scan_prefix unused: the arrays below)
  const double * scans[kSeqMaxScans];
  int32_t prefix[kSeqMaxScans + 1];
  int32_t n_scans, max_n;
  RasterJob * d_job;                       // device copy for the launches that follow
  int32_t clear_blocks;
  long long * dbg;                         // nullptr, or three wall_clock64 stamps of the first scan's workgroup (measurements)
};

// fine pass of the device (results in host-coherent memory)
struct SeqFineOut
{
  int32_t valid;                           // 1 = the device ran the fine pass
  int32_t a, xi, yi;                       // the coarse pass's single best pose
  double centre[3];                        // what the device took as the fine search's centre -- the host checks it against its own
  int32_t bx[4], by[4];                    // ... and the lattice indices it derived
  unsigned long long out[2 + kSeqMaxFine / 2];   // best bits | ties | tie indices (uint32), as in a result block
  int32_t sums[kSeqMaxFine];               // [a][y][x]
};

// what kseq_ties hands to kseq_fine / kseq_done (device memory)
struct SeqMid {int32_t fine, a, bx[3], by[3], pad; double centre[3];};

struct SeqFinalArgs
{
  const uint8_t * job;                     // coarse job (device staging block)
  unsigned long long * h_out;              // host-coherent result block of the coarse pass (lattice part written by kseq_cells)
  SeqFineOut * h_fine;
  int32_t * h_flag; int32_t seq;           // published last, system scope
  int32_t refine, naf, fine_penalize;
  double cx, cy;                           // centre of the coarse search
  const double * xp, * yp;                 // coarse x_poses [nx], y_poses [ny]
  const double * heading;                  // [na]: atan2(sin h, cos h), h = NormalizeAngle(angle a) -- the mean of ONE pose
  const double * fine_cos_sin;             // [na][naf][2]: the fine search's angles around heading[a]
  const double * fine_ang_pen;             // [na][naf]
  const double * fine_dist_pen;            // [9]
  double fxp[3], fyp[3];                   // fine lattice offsets
  int32_t roi_x, roi_y;
  int32_t * fine_table; int32_t * fine_sums;   // the slot's table / volume (introspection reads the last search)
  SeqMid * mid; int32_t * fsum;                // device: the hand-over block, the fine pass's sums while they are added up
  long long * dbg;                         // nullptr, or where the kernel leaves wall_clock64 at its phase boundaries (measurements)
};

void launch_seq_prep(const SeqPrepArgs & args, void * stream);
// The rasteriser's launches, for one job (the descriptor kseq_prep left in device memory) or a batch (blockIdx.y = job): the
// per-job tables travel in the RasterJob (first, cand, seq_ctl, work2)
void launch_seq_prep_batch(const RasterJob * d_jobs, int32_t n_jobs, const ValidItem * d_items, int32_t n_items, int32_t max_n, void * stream);
void launch_seq_links(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, void * stream);
// dynamic LDS of kseq_bin for a job (the host checks it against the device's limit; bm_words = 0: the block map stays in global memory)
size_t seq_bin_lds_bytes(int32_t n_points, int32_t n_foot, int32_t tiles, int32_t bm_words);
// bm_global: the occupancy block map is marked in global memory; dbg (nullptr = none): wall_clock64 at the kernel's phase boundaries
int launch_seq_bin(const RasterJob * d_jobs, int32_t n_jobs, size_t lds_bytes, int32_t bm_global, long long * dbg, void * stream);
size_t seq_tile_table_bytes();
void seq_tile_table(const uint8_t * kernel, int32_t kernel_size, uint8_t * out);
// what the launch between kseq_bin and the scoring of ONE match does besides the stamps: the host's tables (host-coherent memory)
// into device memory, volume and result block of the coarse pass zeroed.  (Always: the first-point table handed back clean.)
struct SeqStageArgs
{
  const void * h_stage; void * d_stage; size_t bytes;
  int32_t * sums; size_t n_sums; unsigned long long * out; size_t out_words;
};
// stamping (smear kernels of >= 8 x 8 cells) + the staging work (stage = nullptr: only the table's reset) in one launch
void launch_seq_tile(const RasterJob * d_jobs, int32_t n_jobs, const uint8_t * d_tab, int32_t max_points, int32_t max_tiles, const SeqStageArgs * stage,
  void * stream);
void launch_raster_tiles(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_tiles, const uint8_t * d_kernel, int32_t kernel_size,
  void * stream);
// ... or on its own, behind the batch path's stamping kernel (smaller smear kernels)
void launch_seq_stage(const RasterJob * d_jobs, int32_t n_jobs, const SeqStageArgs * stage, void * stream);
// (sx = grid cells per lattice step, ry = rows per lane: the windowed kernel's tile shape, pick_ry)
void launch_seq_score(const uint8_t * d_job, int32_t na, int32_t n_points, int32_t nx, int32_t ny, int32_t sx, int32_t ry, void * stream);
void launch_seq_cells(const uint8_t * d_job, int32_t plane, unsigned long long * h_lattice, void * stream);
void launch_seq_final(const SeqFinalArgs & args, int32_t n_points, void * stream);      // kseq_ties, kseq_fine, kseq_done

}  // namespace kh
