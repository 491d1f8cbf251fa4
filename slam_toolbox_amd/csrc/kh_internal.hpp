// Internal definitions shared by the host side (matcher_host.cpp) and the gfx950 kernels
// (matcher_kernels.hip) of libkartohip.  Not part of the public ABI (include/karto_hip.h).
#pragma once
#include <cstddef>
#include <cstdint>

namespace kh
{

constexpr int32_t kInvalidScan = INT32_MAX;   // Math.h:47 INVALID_SCAN
constexpr int32_t kOccupied = 100;            // GridStates_Occupied
constexpr int32_t kTieCap = 2048;             // tie indices returned per CorrelateScan before the host falls back
constexpr int32_t kGridPad = 512;             // zeroed slack before and after the grid so aligned tile reads stay in bounds
constexpr int32_t kTileBytes = 64;            // bytes of one grid row a scoring tile reads (16 lanes x aligned dword)
constexpr int32_t kTileSpan = 61;             // bytes of it that hold poses whatever the alignment class (64 - 3)
constexpr int32_t kClasses = 4;               // alignment classes of a beam offset: (base0 + offset) & 3
constexpr int32_t kBlockShift = 5;             // occupancy block map: one bit per 32 x 32 grid cells ("a stamp touched this block") -- the coarse
                                               // form; handles whose searches are small and whose smear kernel is small keep 8 x 8 blocks
                                               // (RasterJob::bshift / CorrJob::bshift = 3: 9 % fewer windows to score on the config-2 search)
constexpr int32_t kCountsPerAngle = 8;        // counts[a][0..3] = beams per class, [4] = slow beams
// LDS-staged scoring (k_score_lds): a workgroup scores kGroupAngles adjacent angles; the beams are cut into chunks of
// consecutive beams whose windows' union fits one LDS region (built by kLdsRanges waves, a quarter of the beams each)
// (geometry as macros: a measurement build can change it -- four angles on 2 x 74 KB regions of 320 B x 236 rows, one workgroup of
// sixteen waves per compute unit, scored 0.50 ms per launch against 0.44: chunks end at the builder's runs of 64 beams, not at the region)
#ifndef KH_GROUP_ANGLES
#define KH_GROUP_ANGLES 2
#endif
#ifndef KH_LDS_PITCH
#define KH_LDS_PITCH 192
#endif
#ifndef KH_LDS_ROWS
#define KH_LDS_ROWS 196
#endif
#ifndef KH_LDS_REGION_KB
#define KH_LDS_REGION_KB 37
#endif
constexpr int32_t kGroupAngles = KH_GROUP_ANGLES;
constexpr int32_t kLdsRanges = 4;             // builder waves of K2' (each writes its own descriptor list)
// descriptors one builder wave can write: it sees ceil(P / 256) runs of 64 beams, every beam of a run may end up a chunk of its own
inline __host__ __device__ int32_t lds_desc_capacity(int32_t n_points) {return 64 * ((n_points + 255) / 256);}
constexpr int32_t kChunkWords = 8;            // int32 words per chunk descriptor: first beam, grid offset, rows, windows, per angle the class counts
static_assert(4 + kGroupAngles <= kChunkWords, "descriptor words");
constexpr int32_t kLdsPitch = KH_LDS_PITCH;   // bytes per staged grid row (64 mod 128: conflict-free ds_read_b32 of two rows)
constexpr int32_t kLdsRows = KH_LDS_ROWS;     // rows of one staged region
constexpr int32_t kLdsRegionBytes = KH_LDS_REGION_KB * 1024;   // >= kLdsRows * kLdsPitch, a multiple of the 1 KB an LDS-DMA instruction fills; double buffered

// One rasterisation job (ScanMatcher::AddScans, Mapper.cpp:1032-1105) -- device visible.
// The base scans' UNFILTERED point readings live once per distinct scan in the batch's arena; a job names its scans
// (container order).  Job point p = beam p - scan_prefix[k] of base scan k, scan_prefix[k] <= p < scan_prefix[k + 1]:
// the order FindValidPoints / AddScan visit them in.
constexpr int32_t kMaxFootprint = 4;      // kernel cells equal to 100 besides the centre (sigma / res >= 9.9875: the 4-neighbours)
constexpr uint32_t kHashEmpty = 0xffffffffu;
struct RasterJob
{
  uint8_t * grid;            // data_size + kGridPad bytes
  const double * const * scan_ptr;   // n_scans: the readings (x0, y0, x1, y1, ...) of base scan k -- in the batch's upload arena, or
                                     // wherever the caller keeps the scan resident on the device (kh_scan::device_points_xy)
  const int32_t * scan_prefix;   // n_scans + 1: first job point of base scan k
  int32_t n_scans;
  int32_t uniform_n;         // readings per base scan when all the job's scans have the same count (0 = ragged: bisect the prefix)
  double view_x, view_y;     // FindValidPoints' viewpoint = the query scan's sensor position (Mapper.cpp:572)
  uint8_t * active;          // n_points: 1 = the point is stamped.  Written by k_find_valid (FindValidPoints, Mapper.cpp:1113-1164),
                             // narrowed by the order-dependent "cell already occupied" rule (Mapper.cpp:1093-1096) when n_foot > 0
  int32_t n_points;          // sum of the scans' reading counts
  int32_t ws, roi_x, roi_y, roi_w, roi_h;
  int32_t kernel_size;
  double off_x, off_y, scale;   // CoordinateConverter (Karto.h:4421-4436)
  uint32_t * blockmap;       // bm_h rows of bm_w words, one BIT per 32 x 32-cell block (bit bx & 31 of word bx >> 5; the last
                             // word of a row is padding), cleared with the grid: 1 = some stamp's footprint overlaps the block
  int32_t bm_w, bm_h;
  int32_t bshift;            // log2 of the block side in cells (5 or 3)
  // tiled stamping (k_raster_*): kRasterTile x kRasterTile cell tiles, points binned to the <= 2 x 2 tiles
  // their footprint overlaps.  All int32 scratch, zeroed with the grid where noted.
  int32_t tiles_w, tiles_h, height;
  int32_t * tile_count;      // tiles_w * tiles_h   (zeroed) incidences per tile
  int32_t * tile_start;      // tiles_w * tiles_h   first list entry of the tile
  int32_t * tile_cursor;     // tiles_w * tiles_h   (zeroed) fill cursor
  int32_t * work;            // tiles_w * tiles_h   non-empty tiles
  int32_t * n_work;          // 1                   (written by the scan)
  int32_t * cell_xy;         // 2 * n_points        grid cell of every kept point, x < 0 = dropped
  int32_t * rank;            // 4 * n_points        position of the point in the list of each of the <= 4 tiles its footprint overlaps
  int32_t * list;            // 4 * n_points        point indices, tile after tile
  // order-dependent rule (only when the smear kernel holds 100 off-centre): per-job open-addressing table over the ROI
  // cells the valid points fall into
  int32_t n_foot;            // off-centre kernel cells equal to 100 (0 = the rule cannot fire between different cells)
  int32_t foot_dx[kMaxFootprint], foot_dy[kMaxFootprint];
  int32_t hcap;              // slots, a power of two >= 2 * n_points
  uint32_t * hkeys;          // hcap: ROI cell index gy * roi_w + gx, kHashEmpty = free (reset by k_raster_clear)
  int32_t * hvals;           // hcap: smallest job point index that falls into the cell (reset to INT32_MAX)
  uint8_t * hstate;          // hcap: 0 undecided, 1 active, 2 inactive
  int32_t * hnbr;            // hcap * kMaxFootprint: slot of the neighbouring cell's entry, -1 = none
  // re-pitched copies of the grid for the scoring kernel (see CorrJob::grid2), kept in step with the grid tile by tile
  uint8_t * grid2;           // nullptr = this slot has none
  int32_t pitch2, copy_b;
  int32_t copy_kind;         // 1 = copies A / B of the grid itself; 2 = column-decimated copies (CorrJob::dec): pitch2 = their row
                             // pitch, copy_b = bytes of one of the four (even A | odd A | even B | odd B)
  int32_t * prev_work;       // [0] = number of tiles the PREVIOUS rasterisation touched, [4 ...] = their indices
  // first-point rasteriser (matcher_seq.hip: the fused path of one MatchScan, and batches on handles whose tables fit)
  int32_t * first;           // roi_w * roi_h: smallest job point in the cell, INT32_MAX = none (handed back clean by every rasterisation)
  int32_t * cand;            // n_points * 8: stamp candidates (point, cx, cy, stamped | four earlier neighbour points)
  int32_t * seq_ctl;         // 16 control words: [0] candidates
  int32_t * work2;           // 4 * tiles: (tile, list start, count, -) per non-empty tile, for kseq_tile; nullptr = lists hold point indices
};
constexpr int32_t kRasterTile = 64;

// One CorrelateScan job (Mapper.cpp:712-862) -- device visible.  All pointers are device pointers.
struct CorrJob
{
  const uint8_t * grid;
  int32_t data_size, ws;
  int32_t n_points;          // P: beams of the query scan (table row length, response denominator)
  int32_t nx, ny, na;
  int32_t linear;            // lattice base index is base0 + xi*sx + yi*sy_ws for all (xi, yi)
  int32_t sx;                // 1 or 2 when linear (cells per x step)
  int32_t sy_ws;             // bytes per y step when linear
  int32_t base0;             // ROI-shifted grid index of lattice point (0,0) (Mapper.h:1122-1128)
  int32_t tiles_x, tiles_y;  // scoring tiles over the lattice
  int32_t ry;                // rows per lane of the scoring kernel
  int32_t do_penalize;
  int32_t coarse;            // !doingFineMatch: maintain probs (max over angle per (x, y))
  int32_t write_resp;        // also store the penalised response volume (parity tests)
  double denom;              // P * 100 (Mapper.cpp:1204)
  double grid_off_x, grid_off_y, scale;
  // inputs staged by the host (exact libm / reference arithmetic)
  const int32_t * bx;        // nx: gx + roi_x            (Mapper.cpp:660-662)
  const int32_t * by;        // ny: (gy + roi_y) * ws
  const double * dist_pen;   // ny*nx distance penalty     (Mapper.cpp:673-677)
  const double * ang_pen;    // na angle penalty           (Mapper.cpp:679-682)
  const double * cos_sin;    // 2*na: cos, sin of each search angle (Karto.h:6857-6858)
  const double * local;      // 2*P scan points in the sensor frame (Karto.h:6813-6824)
  const uint8_t * invalid;   // P: range reading is NaN/inf (Karto.h:6869-6875)
  // device scratch / outputs
  int32_t * table;           // na*P full lookup table (Karto.h:6844-6894)
  int32_t * fast;            // na*4*lt*P compacted offsets valid for every pose of the lattice, bucketed by alignment
                             // class (base0 + offset) & 3 and scoring tile: list (a, c, t) starts at ((a*4 + c)*lt + t)*P
  int32_t * tcounts;         // na*4*lt list lengths
  int32_t list_tiles;        // lt: tiles_x * tiles_y when <= 32, else 1 (one list per (a, c) shared by all tiles)
  int32_t * slow;            // na*P compacted offsets needing the per-pose range check
  int32_t * counts;          // na*8: {n_class0..3, n_slow, -, -, -}
  // LDS-staged path (lds_path != 0: linear lattice whose window fits 64 bytes x 64 rows)
  int32_t lds_path;
  int32_t sy_cells;          // grid rows per lattice step in y (sy_ws / ws)
  int32_t * rel;             // na*P: byte offset of the beam's window start inside its sub-chunk's LDS region, or -1
  int32_t * chunks;          // [groups][kLdsRanges][lds_desc_capacity(P)][kChunkWords]: chunk descriptors
  int32_t * chunk_counts;    // [groups][kLdsRanges]: chunks of the beam range
  // empty-window skipping: a beam whose whole window lies in blocks no stamp touched adds 0 to every pose
  const uint32_t * blockmap; // see RasterJob; nullptr = do not skip
  int32_t bm_w, bm_h, bshift;
  double * tile_best;        // [na][tiles_y * tiles_x] best response of every scoring tile (K3 -> K4); nullptr = K4 scans everything
  int32_t * sums;            // [na][ny][nx] raw GetResponse numerators (Mapper.cpp:1200)
  double * resp;             // [na][ny][nx] penalised responses (only when write_resp)
  unsigned long long * out;  // result block, see below (zeroed by K2: out_words words)
  int32_t out_words;
  // Re-pitched copies (dual-copy layout).  A wave-level load of K3 reads four 64-byte row segments; in the grid's own
  // pitch (a multiple of 8) a segment straddles a 128-byte cache line almost half of the time, and every straddle is one
  // more L1 tag lookup (measured 6.3 per load, 4 rows x ~1.5 lines).  Copy A holds the same bytes at a pitch that is a
  // multiple of 128, copy B the same again 64 bytes further along every row: for any window, one of the two has every
  // row segment inside ONE line (4 lookups per load).  K2 picks the copy per beam and stores the offset in copy
  // coordinates in a second set of lists; windows that wrap around the row end keep reading the grid itself.
  const uint8_t * grid2;     // copy A; nullptr = none.  copy B = grid2 + copy_b
  int32_t pitch2, copy_b;
  int32_t * fast2;           // like `fast`, offsets into grid2
  int32_t * tcounts2;        // like `tcounts`
  // Column-decimated copies (dec != 0, lattices with sx == 2: the coarse search of MatchScan steps two cells).  Scored from
  // the grid itself, half of every loaded dword belongs to no pose.  Copy E holds the even columns of every row, copy O the
  // odd ones, each at a pitch that is a multiple of 128: a window that starts in column wx0 reads copy (wx0 & 1) from column
  // wx0 >> 1 on, one byte per pose -- the sx == 2 search becomes an sx == 1 search (61 poses per tile row instead of 31) on
  // the copy, and K3 runs its SX = 1 instance.  Both exist twice, the second 64 bytes further along (as above).  The 64
  // bytes behind the ws / 2 columns of row y repeat the start of row y + 1: a window that runs over the row end reads on in
  // the next row like the linear grid index does (Appendix A.3), so every fast beam is served by the copies.
  int32_t dec;
  int32_t pad_rows;          // zero rows in front of and behind the grid and each of its copies
  int32_t pad;               // zero bytes in front of and behind the grid (and, row for row, its copies) that windows may read: a
                             // pose whose index falls off the array adds nothing in the reference (Mapper.cpp:1192-1197) and a
                             // zero here -- beams whose window leaves the array by less than this stay on the fast lists
  int32_t tile_px;           // poses per row of a scoring tile: 61, or 31 for sx == 2 read from the grid itself
  unsigned long long * load_counter;   // handle-wide tally of the row loads K3 will issue for the fast lists K2 builds (wave-level
                                       // dword-load instructions, 256 B each): the L1 side of the roofline; nullptr = not counted
};

// Result block layout (unsigned long long words) per job, zeroed before every CorrelateScan:
//   [0]              best response, IEEE bits (responses are >= 0 so the bit pattern orders like the value)
//   [1]              number of poses within KT_TOLERANCE of the best (Mapper.cpp:808)
//   [2 .. 2+cap/2)   tie indices, uint32, in the reference's order index (y*nX + x)*nA + a
//   [2+cap/2 .. )    probs: nx*ny doubles as bits, max over angle (Mapper.cpp:781-799)
constexpr size_t kOutHeaderWords = 2 + kTieCap / 2;

// one (job, base scan) pair of a batch: k_find_valid walks one scan per lane
struct ValidItem {int32_t job, scan;};
void launch_find_valid(const RasterJob * d_jobs, const ValidItem * d_items, int32_t n_items, int32_t max_n, void * stream);
void launch_active_set(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_cap, void * stream);
void launch_raster(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_points, int32_t max_tiles, const uint8_t * d_kernel, int32_t kernel_size,
  void * stream);
void launch_raster_clear(const RasterJob * d_jobs, int32_t n_jobs, void * stream);
void launch_repitch(const RasterJob * d_jobs, int32_t n_jobs, int32_t max_tiles, void * stream, bool any_copies);          // after launch_raster
void launch_repitch_full(const RasterJob * d_job, int32_t rows, void * stream);                             // one job: whole grid
void launch_offsets(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, void * stream);
void launch_score(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_tiles, int32_t max_na,
                  int32_t sx_variant, int32_t ry, void * stream, bool mfma = false);
// poses per tile row of the scoring kernel for a lattice step of sx cells
// LDS-staged scoring: how many of an angle's four waves share the lattice rows (16 each); the others split the beams.
// Three is rounded up to four: the parts of the steps are dealt by a power-of-two mask.
__host__ __device__ inline int32_t lds_row_waves(int32_t ny) {return ny <= 16 ? 1 : ny <= 32 ? 2 : 4;}
inline int32_t score_tile_poses(int32_t sx) {return sx == 2 ? (kTileSpan + 1) / 2 : kTileSpan;}
void launch_offsets_lds(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, void * stream);
// full_rows: every job's lattice has more than 32 rows (lds_row_waves = 4)
void launch_score_lds(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_na, int32_t sx_variant, bool full_rows, void * stream);
void launch_gather_small(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t * d_out, int32_t small_stride, void * stream);
void launch_ties(const uint8_t * d_jobs, size_t stride, int32_t n_jobs, int32_t max_poses, int32_t tile_pairs, void * stream);

}  // namespace kh
