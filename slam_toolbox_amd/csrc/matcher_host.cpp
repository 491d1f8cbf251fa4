// Host side of the karto correlative scan matcher on MI355X: everything the reference does with
// libm or in inherently sequential order is done here, in the reference's exact IEEE operation
// order (compiled -ffp-contract=off); everything data-parallel is launched on the GPU
// (matcher_kernels.hip).  There is no CPU scoring path.
//
// Reference: lib/karto_sdk/src/Mapper.cpp:477-1208 (ScanMatcher), Mapper.h:1074-1314
// (CorrelationGrid), Karto.h:4393-4563 (CoordinateConverter), :6603-6963 (GridIndexLookup),
// :2946-3041 (Transform), Math.h.
#include "matcher_private.hpp"
#include "matcher_seq.hpp"

namespace kh
{
thread_local std::string g_last_error;
void set_error(const std::string & s) {g_last_error = s;}
}  // namespace kh


namespace kh
{

static int32_t half_kernel_size(double smear, double resolution)   // Mapper.h:1275-1280
{
  return static_cast<int32_t>(round_half_away(2.0 * smear / resolution));
}


// ---- host worker pool: host_pool.hpp ----
void host_parallel_for(size_t n, const std::function<void(size_t)> & fn) {HostPool::instance().run(n, fn);}
void host_parallel_for_wide(size_t n, const std::function<void(size_t)> & fn) {HostPool::wide().run(n, fn);}

// the geometry and scratch pointers of one slot's rasterisation job (the scan list and the order-dependent rule's tables are the
// caller's)
void fill_raster_job(const kh_matcher * m, const Slot & s, const double * pose, int32_t n_points, size_t npad, RasterJob & j)
{
  std::memset(&j, 0, sizeof(j));
  j.grid = s.d_grid;
  j.view_x = pose[0]; j.view_y = pose[1];
  j.active = s.d_ractive; j.n_points = n_points;
  j.ws = m->ws; j.roi_x = m->roi_x; j.roi_y = m->roi_y; j.roi_w = m->roi_w; j.roi_h = m->roi_h;
  j.kernel_size = m->kernel_size; j.off_x = s.off_x; j.off_y = s.off_y; j.scale = m->scale;
  j.blockmap = s.d_blockmap; j.bm_w = m->bm_w; j.bm_h = m->bm_h; j.bshift = m->bshift;
  const size_t nt = static_cast<size_t>(m->rt_w) * m->rt_h;
  j.tiles_w = m->rt_w; j.tiles_h = m->rt_h; j.height = m->data_size / m->ws;
  j.tile_count = s.d_rtiles; j.tile_cursor = s.d_rtiles + nt; j.n_work = s.d_rtiles + 2 * nt;
  j.tile_start = s.d_rtiles + 2 * nt + 4; j.work = s.d_rtiles + 3 * nt + 4;
  j.cell_xy = s.d_rlists; j.list = s.d_rlists + 2 * npad; j.rank = s.d_rlists + 6 * npad;
  j.grid2 = s.d_grid2; j.copy_kind = s.copy_kind; j.prev_work = s.d_prev_work;
  j.pitch2 = s.copy_kind == 2 ? m->pitch_d : m->pitch2; j.copy_b = s.copy_kind == 2 ? m->copy_q : m->copy_b;
  j.n_foot = static_cast<int32_t>(m->footprint100.size()) - 1;
  int32_t f = 0;
  for (const Cell & c : m->footprint100) {
    if (c.x == 0 && c.y == 0) {continue;}
    j.foot_dx[f] = c.x; j.foot_dy[f] = c.y; ++f;
  }
}

constexpr int kNoFirstPointTable = -1000;      // internal: ensure_seq_tables could not allocate a slot's first-point table
int ensure_seq_tables(kh_matcher * m, Slot & s, int32_t n_points, RasterJob & j)
{
  hipStream_t st = m->stream;
  int rc = ensure_device(s.d_cand, s.cap_cand, static_cast<size_t>(std::max(n_points, 1)) * 8, st); if (rc) {return rc;}
  if (!s.d_seqctl) {KH_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_seqctl), sizeof(int32_t) * 16));}
  const bool fused_tiles = m->kernel_size >= 8;
  if (fused_tiles) {
    rc = ensure_device(s.d_work2, s.cap_work2, 4 * static_cast<size_t>(m->rt_w) * m->rt_h, st); if (rc) {return rc;}
    if (!m->d_tab) {
      std::vector<uint8_t> tab(seq_tile_table_bytes());
      seq_tile_table(m->kernel.data(), m->kernel_size, tab.data());
      KH_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_tab), tab.size()));
      KH_HIP(hipMemcpy(m->d_tab, tab.data(), tab.size(), hipMemcpyHostToDevice));
    }
  }
  const size_t roi_cells = static_cast<size_t>(m->roi_w) * m->roi_h;
  if (roi_cells > s.cap_first) {
    if (s.d_first) {KH_HIP(hipStreamSynchronize(st)); KH_HIP(hipFree(s.d_first)); s.d_first = nullptr;}
    // (the tables were budgeted at create, 4 B x cells of the region of interest per slot, but are allocated here: when the device
    // cannot give one -- other handles, other tenants -- this handle goes back to the hash-table rasteriser for good)
    if (hipMalloc(reinterpret_cast<void **>(&s.d_first), roi_cells * sizeof(int32_t)) != hipSuccess) {
      (void)hipGetLastError();
      s.d_first = nullptr; s.cap_first = 0;
      m->table_raster = false;
      return kNoFirstPointTable;
    }
    s.cap_first = roi_cells; s.first_clean = false;
  }
  if (!s.first_clean) {
    KH_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s.d_first), INT32_MAX, roi_cells, st));
    s.first_clean = true;
  }
  j.first = s.d_first; j.cand = s.d_cand; j.seq_ctl = s.d_seqctl; j.work2 = fused_tiles ? s.d_work2 : nullptr;
  return KH_OK;
}

// ---- rasterisation of n jobs (slots[i] <- base scans of job i) ------------------------------

int raster_batch(kh_matcher * m, const std::vector<RasterReq> & reqs)
{
  if (reqs.empty()) {return KH_OK;}
  const double res = m->grid_resolution();
  const size_t n_jobs = reqs.size();
  static const bool timing = std::getenv("KH_MATCH_TIMING") != nullptr;
  const auto t_enter = std::chrono::steady_clock::now();
  // 1. The base scans' unfiltered point readings go to the device ONCE per distinct scan of the batch (loop-closure
  // chains overlap heavily: 256 pairs name ~6400 scans, ~1600 of them distinct); FindValidPoints, WorldToGrid and the
  // order-dependent "cell already occupied" rule all run on the GPU (K0 / K1).  The host only lists who reads what.
  struct Copy {const double * src; size_t dst; int32_t n;};
  std::vector<Copy> copies;
  std::unordered_map<const double *, int32_t> arena_of;        // points_xy -> first point in the arena
  arena_of.reserve(256);
  size_t arena_points = 0, meta_words = 0, n_items = 0;
  std::vector<size_t> meta_at(n_jobs), scans_of(n_jobs, 0);
  std::vector<int32_t> points_of(n_jobs, 0);
  for (size_t r = 0; r < n_jobs; ++r) {
    int64_t pts = 0;
    for (int32_t b = 0; b < reqs[r].n_base; ++b) {
      const kh_scan & sc = reqs[r].base[b];
      if (sc.points_xy == nullptr || sc.n <= 0) {continue;}      // NULL scan: skipped (Mapper.cpp:1039-1041)
      ++scans_of[r]; pts += sc.n;
      if (sc.device_points_xy) {continue;}                        // resident on the device: nothing to upload
      auto it = arena_of.find(sc.points_xy);
      if (it == arena_of.end()) {
        arena_of.emplace(sc.points_xy, static_cast<int32_t>(arena_points));
        copies.push_back(Copy{sc.points_xy, arena_points, sc.n});
        arena_points += static_cast<size_t>(sc.n);
      }
    }
    if (pts > (1 << 30) || arena_points > (1u << 30)) {set_error("too many base scan points in one batch"); return KH_ERR_INVALID_ARG;}
    points_of[r] = static_cast<int32_t>(pts);
    meta_at[r] = meta_words;
    meta_words += 3 * scans_of[r] + 1 + ((scans_of[r] + 1) & 1);      // 64-bit scan pointers, then the prefix (kept 8-byte aligned)
    n_items += scans_of[r];
  }
  const size_t items_at = meta_words;
  // The first-point rasteriser (matcher_seq.hip) for the whole batch: no hash tables, the order-dependent rule and the binning of
  // a job in ONE workgroup with its state in LDS.  Needs every job inside its fixed-size tables.
  const int32_t n_foot_all = static_cast<int32_t>(m->footprint100.size()) - 1;
  bool use_table = m->table_raster;
  size_t bin_lds = 0;
  bool bm_global = false;
  if (use_table) {
    const int32_t tiles = m->rt_w * m->rt_h, bm_words = m->bm_w * m->bm_h;
    int32_t most = 0;
    for (size_t r = 0; r < n_jobs; ++r) {
      most = std::max(most, points_of[r]);
      for (int32_t b = 0; b < reqs[r].n_base; ++b) {use_table = use_table && reqs[r].base[b].n <= kSeqMaxReadings;}
    }
    bin_lds = seq_bin_lds_bytes(most, n_foot_all, tiles, bm_words);
    bm_global = bin_lds > 150 * 1024;
    if (bm_global) {bin_lds = seq_bin_lds_bytes(most, n_foot_all, tiles, 0);}
    use_table = use_table && most <= kSeqMaxPoints && bin_lds <= 150 * 1024;
  }
  meta_words += 2 * n_items;
  int rc = ensure_pinned(m->h_arena, m->cap_harena, std::max<size_t>(arena_points, 1) * 2, m->stream); if (rc) {return rc;}
  rc = ensure_device(m->d_arena, m->cap_darena, std::max<size_t>(arena_points, 1) * 2, m->stream); if (rc) {return rc;}
  rc = ensure_pinned(m->h_meta, m->cap_hmeta, std::max<size_t>(meta_words, 1), m->stream); if (rc) {return rc;}
  rc = ensure_device(m->d_meta, m->cap_dmeta, std::max<size_t>(meta_words, 1), m->stream); if (rc) {return rc;}
  // the pinned mirrors are reused by every call: the previous call's uploads must have left them
  KH_HIP(hipStreamSynchronize(m->stream));
  const int32_t n_foot = static_cast<int32_t>(m->footprint100.size()) - 1;
  int32_t max_points = 0, max_cap = 0, max_scan_n = 1;
  bool any_copies = false;
  ValidItem * items = reinterpret_cast<ValidItem *>(m->h_meta + items_at);
  size_t item = 0;
  for (size_t r = 0; r < n_jobs; ++r) {
    if (r == 0) {max_points = 0; max_cap = 0; max_scan_n = 1; any_copies = false; item = 0;}      // (also on the restart below)
    Slot & s = m->slots[reqs[r].slot];
    const double * pose = reqs[r].query->sensor_pose;
    // MatchScan steps 1-4, Mapper.cpp:543-569
    s.off_x = pose[0] - (0.5 * (m->roi_w - 1) * res);
    s.off_y = pose[1] - (0.5 * (m->roi_h - 1) * res);
    const double ** scan_ptr = reinterpret_cast<const double **>(m->h_meta + meta_at[r]);
    int32_t * scan_prefix = m->h_meta + meta_at[r] + 2 * scans_of[r];
    int32_t k = 0, run = 0, uniform_n = -1;
    for (int32_t b = 0; b < reqs[r].n_base; ++b) {
      const kh_scan & sc = reqs[r].base[b];
      if (sc.points_xy == nullptr || sc.n <= 0) {continue;}
      uniform_n = uniform_n < 0 ? sc.n : (uniform_n == sc.n ? uniform_n : 0);
      max_scan_n = std::max(max_scan_n, sc.n);
      scan_ptr[k] = sc.device_points_xy ? sc.device_points_xy : m->d_arena + 2 * static_cast<size_t>(arena_of.at(sc.points_xy));
      scan_prefix[k] = run;
      run += sc.n;
      items[item].job = static_cast<int32_t>(r); items[item].scan = k; ++item;
      ++k;
    }
    scan_prefix[k] = run;
    const size_t np = static_cast<size_t>(points_of[r]);
    max_points = std::max(max_points, points_of[r]);
    rc = ensure_device(s.d_ractive, s.cap_ractive, std::max<size_t>(np, 1), m->stream); if (rc) {return rc;}
    const size_t npad = (std::max<size_t>(np, 1) + 3) & ~static_cast<size_t>(3);      // the rank quadruples are read as int4
    rc = ensure_device(s.d_rlists, s.cap_rlists, npad * 10, m->stream); if (rc) {return rc;}
    RasterJob & j = m->h_rjobs[r];
    fill_raster_job(m, s, pose, points_of[r], npad, j);
    j.scan_ptr = reinterpret_cast<const double * const *>(m->d_meta + meta_at[r]); j.scan_prefix = m->d_meta + meta_at[r] + 2 * scans_of[r];
    j.n_scans = static_cast<int32_t>(scans_of[r]);
    j.uniform_n = std::max(uniform_n, 0);
    any_copies = any_copies || s.d_grid2 != nullptr;
    j.n_foot = n_foot;
    if (use_table) {
      rc = ensure_seq_tables(m, s, points_of[r], j);
      if (rc == kNoFirstPointTable) {use_table = false; r = static_cast<size_t>(-1); continue;}      // the jobs again, with the hash tables
      if (rc) {return rc;}
      s.first_clean = false;                      // until this batch's stamping launch has handed the table back
    } else if (n_foot > 0) {
      // AddScan's "cell already occupied -> skip" (Mapper.cpp:1093-1096) is order dependent as soon as the smear kernel
      // writes 100 off-centre: cell table for k_cell_first / k_active_set
      size_t cap = 1024;
      while (cap < 2 * std::max<size_t>(np, 1)) {cap <<= 1;}
      rc = ensure_device(s.d_hkeys, s.cap_hkeys, cap, m->stream); if (rc) {return rc;}
      rc = ensure_device(s.d_hvals, s.cap_hvals, cap, m->stream); if (rc) {return rc;}
      rc = ensure_device(s.d_hstate, s.cap_hstate, cap, m->stream); if (rc) {return rc;}
      rc = ensure_device(s.d_hnbr, s.cap_hnbr, cap * kMaxFootprint, m->stream); if (rc) {return rc;}
      j.hcap = static_cast<int32_t>(cap);
      max_cap = std::max(max_cap, j.hcap);
      j.hkeys = s.d_hkeys; j.hvals = s.d_hvals; j.hstate = s.d_hstate; j.hnbr = s.d_hnbr;
      int32_t f = 0;
      for (const Cell & c : m->footprint100) {
        if (c.x == 0 && c.y == 0) {continue;}
        j.foot_dx[f] = c.x; j.foot_dy[f] = c.y; ++f;
      }
    }
  }
  // 2. upload the jobs and scan lists; Grid::Clear (Karto.h:4612-4615) -- up to 16.8 MB per job -- needs nothing else and
  // runs while the pool gathers the distinct scans' points into the pinned arena
  if (meta_words) {
    KH_HIP(hipMemcpyAsync(m->d_meta, m->h_meta, sizeof(int32_t) * meta_words, hipMemcpyHostToDevice, m->stream));
  }
  KH_HIP(hipMemcpyAsync(m->d_rjobs, m->h_rjobs, sizeof(RasterJob) * n_jobs, hipMemcpyHostToDevice, m->stream));
  if (m->profiling) {KH_HIP(hipEventRecord(m->ev[2], m->stream));}
  if (!use_table) {launch_raster_clear(m->d_rjobs, static_cast<int32_t>(n_jobs), m->stream);}
  HostPool::instance().run(copies.size(), [&](size_t i) {
    std::memcpy(m->h_arena + 2 * copies[i].dst, copies[i].src, sizeof(double) * 2 * static_cast<size_t>(copies[i].n));
  });
  if (arena_points) {
    KH_HIP(hipMemcpyAsync(m->d_arena, m->h_arena, sizeof(double) * 2 * arena_points, hipMemcpyHostToDevice, m->stream));
  }
  // 3. FindValidPoints, stamps
  if (use_table) {
    const int32_t nj = static_cast<int32_t>(n_jobs), tiles = m->rt_w * m->rt_h;
    launch_seq_prep_batch(m->d_rjobs, nj, reinterpret_cast<const ValidItem *>(m->d_meta + items_at), static_cast<int32_t>(n_items), max_scan_n, m->stream);
    launch_seq_links(m->d_rjobs, nj, max_points, m->stream);
    launch_seq_bin(m->d_rjobs, nj, bin_lds, bm_global ? 1 : 0, nullptr, m->stream);
    if (m->kernel_size >= 8) {
      launch_seq_tile(m->d_rjobs, nj, m->d_tab, max_points, tiles, nullptr, m->stream);
    } else {
      launch_raster_tiles(m->d_rjobs, nj, max_points, tiles, m->d_kernel, m->kernel_size, m->stream);
      launch_seq_stage(m->d_rjobs, nj, nullptr, m->stream);
    }
    for (size_t r = 0; r < n_jobs; ++r) {m->slots[reqs[r].slot].first_clean = true;}
  } else {
    launch_find_valid(m->d_rjobs, reinterpret_cast<const ValidItem *>(m->d_meta + items_at), static_cast<int32_t>(n_items), max_scan_n, m->stream);
    if (n_foot > 0) {launch_active_set(m->d_rjobs, static_cast<int32_t>(n_jobs), max_points, max_cap, m->stream);}
    launch_raster(m->d_rjobs, static_cast<int32_t>(n_jobs), max_points, m->rt_w * m->rt_h, m->d_kernel, m->kernel_size, m->stream);
  }
  launch_repitch(m->d_rjobs, static_cast<int32_t>(n_jobs), m->rt_w * m->rt_h, m->stream, any_copies);
  KH_HIP(hipGetLastError());
  if (timing) {
    std::fprintf(stderr, "[kh raster] host %.3f ms: %zu jobs, %zu (job, scan) items, %zu distinct scans, %.1f MB of points uploaded\n",
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count(), n_jobs, n_items, copies.size(),
      arena_points * 16.0 / 1e6);
  }
  if (m->profiling) {
    KH_HIP(hipEventRecord(m->ev[3], m->stream));
    KH_HIP(hipEventSynchronize(m->ev[3]));
    float ms = 0;
    KH_HIP(hipEventElapsedTime(&ms, m->ev[2], m->ev[3]));
    m->raster_ms += ms; m->raster_launches += 1;
  }
  return KH_OK;
}

// ---- CorrelateScan on a set of slots ---------------------------------------------------------

static inline double q_res_x(const CorrReq & q) {return q.res_x;}

// Re-pitched copies of a slot's grid (CorrJob::grid2): allocated zeroed, filled from the grid as it stands, kept in step by
// raster_batch from then on.
static int allocate_copies(kh_matcher * m, Slot & s, int32_t kind)
{
  const size_t rows = static_cast<size_t>(m->data_size / m->ws);
  const size_t bytes = (kind == 2 ? static_cast<size_t>(m->copy_q) * 4 + 64 : static_cast<size_t>(m->copy_b) * 2 + 64) + 2 * kGridPad;
  s.copy_kind = kind;
  KH_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_grid2_alloc), bytes));
  KH_HIP(hipMemsetAsync(s.d_grid2_alloc, 0, bytes, m->stream));
  s.d_grid2 = s.d_grid2_alloc + kGridPad + static_cast<size_t>(m->pad_rows) * (kind == 2 ? m->pitch_d : m->pitch2);
  const size_t nt = static_cast<size_t>(m->rt_w) * m->rt_h;
  // one-job descriptor for the full copy: only the fields k_repitch_full / k_repitch_keep read
  RasterJob j;
  std::memset(&j, 0, sizeof(j));
  j.grid = s.d_grid; j.ws = m->ws; j.height = static_cast<int32_t>(rows);
  j.grid2 = s.d_grid2; j.copy_kind = kind; j.prev_work = s.d_prev_work;
  j.pitch2 = kind == 2 ? m->pitch_d : m->pitch2; j.copy_b = kind == 2 ? m->copy_q : m->copy_b;
  const size_t nt2 = nt;
  j.n_work = s.d_rtiles + 2 * nt2; j.work = s.d_rtiles + 3 * nt2 + 4;
  RasterJob * d_j = nullptr;
  KH_HIP(hipMalloc(reinterpret_cast<void **>(&d_j), sizeof(RasterJob)));
  KH_HIP(hipMemcpyAsync(d_j, &j, sizeof(RasterJob), hipMemcpyHostToDevice, m->stream));
  launch_repitch_full(d_j, static_cast<int32_t>(rows), m->stream);
  KH_HIP(hipStreamSynchronize(m->stream));       // `j` is on this stack, and the side streams may read the copies next
  KH_HIP(hipFree(d_j));
  return KH_OK;
}

// (the distance penalties -- nx * ny doubles, 52 KB of a loop-closure search's 70 -- are only staged for searches that penalise)
StageLayout stage_layout(int32_t P, int32_t nx, int32_t ny, int32_t na, bool penalize)
{
  StageLayout L;
  size_t o = align_up(sizeof(CorrJob), 256);
  L.bx = o; o = align_up(o + sizeof(int32_t) * nx, 16);
  L.by = o; o = align_up(o + sizeof(int32_t) * ny, 16);
  L.dist_pen = o; o = align_up(o + (penalize ? sizeof(double) * nx * ny : 0), 16);
  L.ang_pen = o; o = align_up(o + sizeof(double) * na, 16);
  L.cos_sin = o; o = align_up(o + sizeof(double) * 2 * na, 16);
  L.local = o; o = align_up(o + sizeof(double) * 2 * P, 16);
  L.invalid = o; o = align_up(o + P, 16);
  L.total = align_up(o, 256);
  return L;
}

// rows per lane of the scoring kernel (a tile is 4 * ry lattice rows): the variant that loads the fewest rows for ny --
// 81 rows are 3 tiles of 28 (ry 7: 21 row loads per beam and column of tiles) rather than 3 tiles of 32 (ry 8: 24)
int pick_ry(int32_t ny)
{
  if (ny <= 4) {return 1;}
  int best = 8, best_cost = 1 << 30;
  for (int ry : {8, 7, 4}) {
    const int cost = ((ny + 4 * ry - 1) / (4 * ry)) * ry;
    if (cost < best_cost) {best = ry; best_cost = cost;}
  }
  return best;
}

// Host half of ComputePositionalCovariance (Mapper.cpp:874-966) on the lattice maxima
static int positional_covariance(
  const kh_matcher * m, const CorrHost & c, const std::vector<double> & lattice_max, const WalkGeometry & w,
  const double best_pose[3], double best_response, double * cov)
{
  std::fill(cov, cov + 9, 0.0);
  cov[0] = 1.0; cov[4] = 1.0; cov[8] = 1.0;       // SetToIdentity
  if (best_response < kTolerance) {
    cov[0] = kMaxVariance; cov[4] = kMaxVariance; cov[8] = 4 * (w.ang_res * w.ang_res);
    return KH_OK;
  }
  // search-space-probs grid (Grid<kt_double>, side x side, Mapper.cpp:513-514, 726-732, 781-799)
  const int32_t side = m->side;
  const double pscale = 1.0 / m->resolution;            // Grid::CreateGrid -> SetScale(1.0 / resolution)
  const double pox = c.center[0] - c.off_x, poy = c.center[1] - c.off_y;
  std::vector<double> probs(static_cast<size_t>(side) * side, 0.0);
  for (int32_t yi = 0; yi < c.ny; ++yi) {
    for (int32_t xi = 0; xi < c.nx; ++xi) {
      const double px = c.center[0] + c.x_poses[xi], py = c.center[1] + c.y_poses[yi];
      const Cell g = world_to_grid(pscale, pox, poy, px, py);
      if (!(g.x >= 0 && g.x < side) || !(g.y >= 0 && g.y < side)) {return KH_ERR_SEARCH;}   // Mapper.cpp:786-796
      double & cell = probs[static_cast<size_t>(g.y) * side + g.x];
      const double v = lattice_max[static_cast<size_t>(yi) * c.nx + xi];
      cell = v > cell ? v : cell;
    }
  }
  double aXX = 0, aXY = 0, aYY = 0, norm = 0;
  // the walk uses the CALLER's geometry on the grid the last coarse search left behind (Mapper.cpp:896-923)
  const double dx = best_pose[0] - w.center[0], dy = best_pose[1] - w.center[1];
  const uint32_t nX = static_cast<uint32_t>(round_half_away(w.off_x * 2.0 / w.res_x) + 1);
  const double startX = -w.off_x;
  const uint32_t nY = static_cast<uint32_t>(round_half_away(w.off_y * 2.0 / w.res_y) + 1);
  const double startY = -w.off_y;
  for (uint32_t yi = 0; yi < nY; ++yi) {
    const double y = startY + yi * w.res_y;
    for (uint32_t xi = 0; xi < nX; ++xi) {
      const double x = startX + xi * w.res_x;
      const Cell g = world_to_grid(pscale, pox, poy, w.center[0] + x, w.center[1] + y);
      if (!(g.x >= 0 && g.x < side) || !(g.y >= 0 && g.y < side)) {return KH_ERR_SEARCH;}
      const double response = probs[static_cast<size_t>(g.y) * side + g.x];
      if (response >= (best_response - 0.1)) {
        norm += response;
        aXX += ((x - dx) * (x - dx) * response);
        aXY += ((x - dx) * (y - dy) * response);
        aYY += ((y - dy) * (y - dy) * response);
      }
    }
  }
  if (norm > kTolerance) {
    double vXX = aXX / norm, vXY = aXY / norm, vYY = aYY / norm;
    const double vTHTH = 4 * (w.ang_res * w.ang_res);
    const double minXX = 0.1 * (w.res_x * w.res_x), minYY = 0.1 * (w.res_y * w.res_y);
    vXX = vXX > minXX ? vXX : minXX;
    vYY = vYY > minYY ? vYY : minYY;
    const double mult = 1.0 / best_response;
    cov[0] = vXX * mult; cov[1] = vXY * mult; cov[3] = vXY * mult; cov[4] = vYY * mult; cov[8] = vTHTH;
  }
  if (double_equal(cov[0], 0.0)) {cov[0] = kMaxVariance;}
  if (double_equal(cov[4], 0.0)) {cov[4] = kMaxVariance;}
  return KH_OK;
}

// the accumulation of ComputeAngularCovariance (Mapper.cpp:992-1024) over the raw responses of all angles at the best cell
static double angular_variance(const std::vector<int32_t> & sums, double denom, double center_heading, double ang_off,
  double ang_res, double best_angle, double best_response)
{
  const double startAngle = center_heading - ang_off;
  double norm = 0.0, acc = 0.0;
  for (size_t a = 0; a < sums.size(); ++a) {
    const double angle = startAngle + static_cast<uint32_t>(a) * ang_res;
    const double response = static_cast<double>(sums[a]) / denom;     // GetResponse: no penalty
    if (response >= (best_response - 0.1)) {
      norm += response;
      acc += ((angle - best_angle) * (angle - best_angle) * response);
    }
  }
  if (norm > kTolerance) {
    if (acc < kTolerance) {acc = ang_res * ang_res;}
    acc /= norm;
  } else {
    acc = 1000 * (ang_res * ang_res);
  }
  return acc;
}

static inline double host_response(const CorrHost & c, int32_t sum, int a, int yi, int xi)
{
  double response = static_cast<double>(sum) / c.denom;
  if (c.penalize && !double_equal(response, 0.0)) {
    response *= (c.dist_pen[static_cast<size_t>(yi) * c.nx + xi] * c.ang_pen[a]);
  }
  return response;
}

// the search lattice of one request (Mapper.cpp:736-756)
int init_ctx(const CorrReq & q, CorrHost & c)
{
  c.slot = q.slot;
  c.P = q.scan->n;
  std::copy(q.center, q.center + 3, c.center);
  c.off_x = q.off_x; c.off_y = q.off_y; c.res_x = q.res_x; c.res_y = q.res_y;
  c.ang_off = q.ang_off; c.ang_res = q.ang_res; c.fine = q.fine; c.penalize = q.penalize;
  c.nx = static_cast<int32_t>(static_cast<uint32_t>(round_half_away(q.off_x * 2.0 / q.res_x) + 1));
  c.ny = static_cast<int32_t>(static_cast<uint32_t>(round_half_away(q.off_y * 2.0 / q.res_y) + 1));
  c.na = static_cast<int32_t>(static_cast<uint32_t>(round_half_away(q.ang_off * 2.0 / q.ang_res) + 1));
  if (c.nx <= 0 || c.ny <= 0 || c.na <= 0 || static_cast<int64_t>(c.nx) * c.ny * c.na > (1ll << 28)) {
    set_error("search volume out of range");
    return KH_ERR_INVALID_ARG;
  }
  const double startX = -q.off_x, startY = -q.off_y;
  c.x_poses.resize(c.nx); c.y_poses.resize(c.ny);
  for (int32_t k = 0; k < c.nx; ++k) {c.x_poses[k] = startX + static_cast<uint32_t>(k) * q.res_x;}
  for (int32_t k = 0; k < c.ny; ++k) {c.y_poses[k] = startY + static_cast<uint32_t>(k) * q.res_y;}
  c.denom = static_cast<double>(static_cast<uint32_t>(c.P) * 100u);     // Mapper.cpp:1204
  return KH_OK;
}

// device scratch of one slot for the search `c` describes (allocation is serial: called before the pool fills the tables)
int ensure_slot_scratch(kh_matcher * m, const CorrReq & q, CorrHost & c, bool allow_copies)
{
  (void)q;
  int rc = KH_OK;
  Slot & s = m->slots[c.slot];
  // device scratch for this slot
  const size_t tp = static_cast<size_t>(c.na) * c.P;
  if (tp > s.cap_table) {
    if (s.d_table) {KH_HIP(hipStreamSynchronize(m->stream)); KH_HIP(hipFree(s.d_table)); KH_HIP(hipFree(s.d_slow));}
    const size_t cap = std::max(tp, s.cap_table + s.cap_table / 2);
    KH_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_table), cap * 4));
    KH_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_slow), cap * 4));
    s.cap_table = cap;
  }
  rc = ensure_device(s.d_counts, s.cap_counts, static_cast<size_t>(c.na) * kCountsPerAngle, m->stream); if (rc) {return rc;}
  {
    // one compacted list per (angle, alignment class, scoring tile): bound of the tile count, the job's real
    // one (<= it) is set with the rest of the job below
    size_t lt = static_cast<size_t>((c.nx + 30) / 31) * ((c.ny + 4 * pick_ry(c.ny) - 1) / (4 * pick_ry(c.ny)));
    if (lt > 32) {lt = 1;}
    c.lt_alloc = static_cast<int32_t>(lt);
    // (twice: the second half holds the lists into the re-pitched copies)
    rc = ensure_device(s.d_fast, s.cap_fast, 2 * tp * kClasses * lt, m->stream); if (rc) {return rc;}
    rc = ensure_device(s.d_tcounts, s.cap_tcounts, 2 * static_cast<size_t>(c.na) * kClasses * lt, m->stream); if (rc) {return rc;}
    // dual-copy layout: worth its memory (2 x the grid) and upkeep for full-resolution searches with many angles whose
    // window is one tile wide -- the config-2 CorrelateScan; decided from the request alone, allocated once per slot
    const double work = static_cast<double>(c.nx) * c.ny * c.na * c.P;
    // (the copy is picked per beam and scoring tile, so the lattice is one tile wide or has per-tile lists)
    const double cells_per_step = q.res_x * m->scale;
    const bool full_res = c.nx > 1 && std::fabs(cells_per_step - 1.0) < 1e-9 && c.nx <= kTileSpan;
    const bool tiled_lists = lt > 1 && (std::fabs(cells_per_step - 1.0) < 1e-9 || std::fabs(cells_per_step - 2.0) < 1e-9);
    if (allow_copies && m->dual_copy && !s.d_grid2 && (full_res || tiled_lists) && work >= 1e8) {
      // a search that steps two cells (MatchScan's coarse pass) gets the column-decimated copies
      const bool two_cells = std::fabs(cells_per_step - 2.0) < 1e-9 && m->copy_q > 0 && (m->ws % 8) == 0;
      rc = allocate_copies(m, s, two_cells ? 2 : 1); if (rc) {return rc;}
    }
  }
  {
    // chunk descriptors of the LDS-staged path: at most one per beam, kept per (angle pair, beam range)
    const size_t groups = (static_cast<size_t>(c.na) + kGroupAngles - 1) / kGroupAngles;
    const size_t range_len = static_cast<size_t>(lds_desc_capacity(c.P));
    rc = ensure_device(s.d_chunks, s.cap_chunks, groups * kLdsRanges * range_len * kChunkWords, m->stream); if (rc) {return rc;}
    rc = ensure_device(s.d_chunk_counts, s.cap_chunk_counts, groups * kLdsRanges, m->stream); if (rc) {return rc;}
  }
  const size_t vol = static_cast<size_t>(c.nx) * c.ny * c.na;
  rc = ensure_device(s.d_sums, s.cap_volume, vol, m->stream); if (rc) {return rc;}
  // best response per scoring tile: at most ceil(nx / 31) x ceil(ny / 4) tiles per angle
  rc = ensure_device(s.d_tile_best, s.cap_tile_best,
      static_cast<size_t>(c.na) * ((c.nx + 30) / 31) * ((c.ny + 3) / 4), m->stream); if (rc) {return rc;}
  if (m->keep_responses) {rc = ensure_device(s.d_resp, s.cap_resp, vol, m->stream); if (rc) {return rc;}}

  return KH_OK;
}

// Host half of one CorrelateScan job: the exact-arithmetic tables into the staging block `hb` (device address `db`) and the job
// descriptor at its head.  Independent between the jobs of a batch (runs on the pool).
void prepare_job(kh_matcher * m, const CorrReq & q, CorrHost & c, const StageLayout & L, uint8_t * hb, uint8_t * db,
  unsigned long long * d_out, size_t out_words, size_t n_launch, bool lds_always, bool lds_never, JobShape & shape)
{
  const kh_match_params & mp = m->params;
  Slot & s = m->slots[c.slot];
  CorrJob * job = reinterpret_cast<CorrJob *>(hb);
  int32_t * bx = reinterpret_cast<int32_t *>(hb + L.bx);
  int32_t * by = reinterpret_cast<int32_t *>(hb + L.by);
  double * dist_pen = reinterpret_cast<double *>(hb + L.dist_pen);
  double * ang_pen = reinterpret_cast<double *>(hb + L.ang_pen);
  double * cos_sin = reinterpret_cast<double *>(hb + L.cos_sin);
  double * local = reinterpret_cast<double *>(hb + L.local);
  uint8_t * invalid = hb + L.invalid;

  // lattice base indices: operator()(y), Mapper.cpp:649-662
  c.bx.resize(c.nx); c.by.resize(c.ny);
  for (int32_t k = 0; k < c.nx; ++k) {
    const double newPositionX = c.center[0] + c.x_poses[k];
    const double gx = (newPositionX - s.off_x) * m->scale;
    c.bx[k] = to_int32(round_half_away(gx)) + m->roi_x;
    bx[k] = c.bx[k];
  }
  for (int32_t k = 0; k < c.ny; ++k) {
    const double newPositionY = c.center[1] + c.y_poses[k];
    const double gy = (newPositionY - s.off_y) * m->scale;
    c.by[k] = (to_int32(round_half_away(gy)) + m->roi_y) * m->ws;
    by[k] = c.by[k];
  }
  int32_t sx = c.nx > 1 ? c.bx[1] - c.bx[0] : 1;
  int32_t sy_ws = c.ny > 1 ? c.by[1] - c.by[0] : m->ws;
  bool linear = (sx == 1 || sx == 2) && sy_ws > 0;
  for (int32_t k = 1; k < c.nx && linear; ++k) {linear = (c.bx[k] - c.bx[k - 1]) == sx;}
  for (int32_t k = 1; k < c.ny && linear; ++k) {linear = (c.by[k] - c.by[k - 1]) == sy_ws;}
  // every window row read by a tile must stay inside the (padded) allocation
  if (linear) {
    const int64_t bmax = static_cast<int64_t>(c.bx[0]) + c.by[0] + static_cast<int64_t>(c.ny - 1) * sy_ws;
    if (c.bx[0] + c.by[0] < 0 || bmax >= m->data_size) {linear = false;}
  }
  if (!linear) {sx = 1; sy_ws = m->ws;}

  // penalties, Mapper.cpp:671-685
  if (q.penalize) {c.dist_pen.assign(static_cast<size_t>(c.nx) * c.ny, 1.0);} else {c.dist_pen.clear();}
  c.ang_pen.assign(c.na, 1.0);
  c.angles.resize(c.na);
  const double startAngle = c.center[2] - c.ang_off;
  for (int32_t a = 0; a < c.na; ++a) {
    const double angle = startAngle + static_cast<uint32_t>(a) * c.ang_res;
    c.angles[a] = angle;
    ref_sincos(angle, &cos_sin[2 * a + 1], &cos_sin[2 * a]);          // Karto.h:6857-6858
    const double squaredAngleDistance = (angle - c.center[2]) * (angle - c.center[2]);
    double anglePenalty = 1.0 - (kAngleGain * squaredAngleDistance / mp.angle_variance_penalty);
    anglePenalty = anglePenalty > mp.minimum_angle_penalty ? anglePenalty : mp.minimum_angle_penalty;
    c.ang_pen[a] = anglePenalty;
    ang_pen[a] = anglePenalty;
  }
  if (q.penalize) {
  for (int32_t yi = 0; yi < c.ny; ++yi) {
    const double squareY = c.y_poses[yi] * c.y_poses[yi];
    for (int32_t xi = 0; xi < c.nx; ++xi) {
      const double squareX = c.x_poses[xi] * c.x_poses[xi];
      const double squaredDistance = squareX + squareY;
      double distancePenalty = 1.0 - (kDistanceGain * squaredDistance / mp.distance_variance_penalty);
      distancePenalty = distancePenalty > mp.minimum_distance_penalty ? distancePenalty : mp.minimum_distance_penalty;
      c.dist_pen[static_cast<size_t>(yi) * c.nx + xi] = distancePenalty;
      dist_pen[static_cast<size_t>(yi) * c.nx + xi] = distancePenalty;
    }
  }
  }

  // scan points in the sensor frame: Transform(sensorPose).InverseTransformPose, Karto.h:6813-6824,
  // 2987-2994, 3003-3024, 2482-2511, 2654-2666
  {
    const double tx = q.scan->sensor_pose[0], ty = q.scan->sensor_pose[1], th = q.scan->sensor_pose[2];
    double r00, r01, r02, r10, r11, r12;
    if (tx == 0.0 && ty == 0.0 && th == 0.0) {
      r00 = 1; r01 = 0; r02 = 0; r10 = 0; r11 = 1; r12 = 0;
    } else {
      const double radians = 0.0 - th;
      double cosR, sinR;
      ref_sincos(radians, &sinR, &cosR);
      const double omc = 1.0 - cosR;
      r00 = 0.0 * omc + cosR;
      r01 = 0.0 * 0.0 * omc - 1.0 * sinR;
      r02 = 0.0 * 1.0 * omc + 0.0 * sinR;
      r10 = 0.0 * 0.0 * omc + 1.0 * sinR;
      r11 = 0.0 * omc + cosR;
      r12 = 0.0 * 1.0 * omc - 0.0 * sinR;
    }
    for (int32_t k = 0; k < c.P; ++k) {
      const double sxp = q.scan->points_xy[2 * k] - tx, syp = q.scan->points_xy[2 * k + 1] - ty, sh = 0.0 - th;
      local[2 * k] = r00 * sxp + r01 * syp + r02 * sh;
      local[2 * k + 1] = r10 * sxp + r11 * syp + r12 * sh;
      const double rr = q.scan->ranges[k];
      invalid[k] = (std::isnan(rr) || std::isinf(rr)) ? 1 : 0;     // Karto.h:6869-6875
    }
  }

  std::memset(job, 0, sizeof(CorrJob));
  job->grid = s.d_grid; job->data_size = m->data_size; job->ws = m->ws;
  job->n_points = c.P; job->nx = c.nx; job->ny = c.ny; job->na = c.na;
  job->linear = linear ? 1 : 0; job->sx = sx; job->sy_ws = sy_ws; job->base0 = c.bx[0] + c.by[0];
  const int this_sx = (linear && sx == 2) ? 2 : 1;
  const int this_ry = pick_ry(c.ny);
  job->tiles_y = (c.ny + 4 * this_ry - 1) / (4 * this_ry);
  // column-decimated copies of the slot: the two-cell search is scored as a one-cell search on them (CorrJob::dec), as long
  // as the copy can be picked per beam and tile (one tile column, or lists per tile)
  bool dec = s.copy_kind == 2 && m->dual_copy && this_sx == 2 && sy_ws % m->ws == 0;
  if (dec) {
    const int32_t tiles = ((c.nx + kTileSpan - 1) / kTileSpan) * job->tiles_y;
    dec = c.nx <= kTileSpan || (tiles >= 4 && tiles <= c.lt_alloc);
  }
  const int px = dec ? kTileSpan : score_tile_poses(this_sx);
  job->tiles_x = (c.nx + px - 1) / px;
  job->ry = this_ry; job->tile_px = px; job->dec = dec ? 1 : 0;
  shape.sx = dec ? 1 : this_sx; shape.ry = this_ry; shape.tiles = job->tiles_x * job->tiles_y;
  // LDS-staged scoring: linear lattice whose window fits 64 bytes x 64 rows
  // ... and a launch of at least two workgroups (angle pairs) per compute unit: one config-2 search alone is 41 workgroups that
  // walk their 25 chunks one after the other -- 0.26 ms against the windowed kernel's 0.14
  const bool lds_wanted = lds_always || (!lds_never && static_cast<double>(c.nx) * c.ny * c.na * c.P >= 1e8 &&
    static_cast<double>(n_launch) * c.na >= 1024.0);      // (workgroup = kGroupAngles angles)
  const bool lds_ok = lds_wanted && linear && (c.nx - 1) * sx + 1 <= kTileSpan && c.ny <= 64 && c.P <= 2048 &&
    sy_ws % m->ws == 0 && sy_ws / m->ws == sx && (m->ws % 4) == 0 && 63 * sx + 1 <= kLdsRows;
  shape.lds = lds_ok ? 1 : 0;
  job->lds_path = shape.lds; job->sy_cells = sy_ws / m->ws;
  job->rel = s.d_fast; job->chunks = s.d_chunks; job->chunk_counts = s.d_chunk_counts;
  job->do_penalize = q.penalize ? 1 : 0; job->coarse = q.fine ? 0 : 1;
  job->write_resp = m->keep_responses ? 1 : 0;
  job->denom = c.denom;
  job->grid_off_x = s.off_x; job->grid_off_y = s.off_y; job->scale = m->scale;
  job->bx = reinterpret_cast<const int32_t *>(db + L.bx);
  job->by = reinterpret_cast<const int32_t *>(db + L.by);
  job->dist_pen = reinterpret_cast<const double *>(db + L.dist_pen);
  job->ang_pen = reinterpret_cast<const double *>(db + L.ang_pen);
  job->cos_sin = reinterpret_cast<const double *>(db + L.cos_sin);
  job->local = reinterpret_cast<const double *>(db + L.local);
  job->invalid = db + L.invalid;
  job->table = s.d_table; job->fast = s.d_fast; job->slow = s.d_slow; job->counts = s.d_counts;
  job->tcounts = s.d_tcounts; {
    // tile lists pay for themselves when the window is several tiles large (the tests cost K2 time per tile)
    const int32_t tiles = job->tiles_x * job->tiles_y;
    job->list_tiles = (tiles >= 4 && tiles <= c.lt_alloc) ? tiles : 1;
  }
  job->sums = s.d_sums; job->resp = s.d_resp; job->out = d_out; job->out_words = static_cast<int32_t>(out_words);
  job->blockmap = m->dense_score ? nullptr : s.d_blockmap; job->bm_w = m->bm_w; job->bm_h = m->bm_h; job->bshift = m->bshift;
  job->tile_best = s.d_tile_best;
  // re-pitched copies: linear full-resolution lattice one tile wide (the copy is picked per beam for the tile at x0 = 0)
  {
    const bool use2 = dec || (s.copy_kind == 1 && m->dual_copy && linear && (job->tiles_x == 1 || job->list_tiles > 1));
    const size_t lists = static_cast<size_t>(c.na) * kClasses * static_cast<size_t>(c.lt_alloc);
    job->grid2 = use2 ? s.d_grid2 : nullptr;
    job->pitch2 = dec ? m->pitch_d : m->pitch2; job->copy_b = dec ? m->copy_q : m->copy_b;
    job->fast2 = s.d_fast + lists * static_cast<size_t>(c.P);
    job->tcounts2 = s.d_tcounts + lists;
  }
  job->pad = std::max(0, m->pad_rows * m->ws - 512); job->pad_rows = m->pad_rows;
  job->load_counter = m->profiling ? m->d_load_counter : nullptr;
}

// Finalisation of one CorrelateScan (Mapper.cpp:775-862) from its downloaded result block: tie average, covariance, the slot's
// introspection state.  q.status carries the reference's "unable to find best position".
int finalize_job(kh_matcher * m, CorrReq & q, CorrHost & c, const ResultView & v)
{
  int rc = KH_OK;
  Slot & s = m->slots[c.slot];
  s.volume_stale = false;
  const unsigned long long * out = v.out;
  double best;
  std::memcpy(&best, &out[0], 8);
  const uint64_t tie_count = out[1];
  const size_t plane = static_cast<size_t>(c.nx) * c.ny;
  std::vector<uint32_t> ties;
  std::vector<int32_t> host_sums;     // full volume, only when needed
  auto fetch_volume = [&]() -> int {
    if (!host_sums.empty()) {return KH_OK;}
    if (!v.device_work) {return kNeedGeneric;}
    host_sums.resize(plane * c.na);
    KH_HIP(hipMemcpy(host_sums.data(), s.d_sums, host_sums.size() * 4, hipMemcpyDeviceToHost));
    return KH_OK;
  };
  if (tie_count <= static_cast<uint64_t>(kTieCap)) {
    const uint32_t * idx = reinterpret_cast<const uint32_t *>(out + 2);
    ties.assign(idx, idx + tie_count);
    std::sort(ties.begin(), ties.end());
  } else {
    // degenerate search (e.g. nothing rasterised: every pose ties at 0): walk the whole volume
    // on the host in the reference's order
    rc = fetch_volume(); if (rc) {return rc;}
    for (int32_t yi = 0; yi < c.ny; ++yi) {
      for (int32_t xi = 0; xi < c.nx; ++xi) {
        for (int32_t a = 0; a < c.na; ++a) {
          const double r = host_response(c, host_sums[static_cast<size_t>(a) * plane + static_cast<size_t>(yi) * c.nx + xi], a, yi, xi);
          if (double_equal(r, best)) {ties.push_back(static_cast<uint32_t>((static_cast<size_t>(yi) * c.nx + xi) * c.na + a));}
        }
      }
    }
  }
  if (ties.empty()) {q.status = KH_ERR_SEARCH; return KH_OK;}     // Mapper.cpp:828
  // average all poses with the same highest response, Mapper.cpp:802-829
  double ax = 0.0, ay = 0.0, thetaX = 0.0, thetaY = 0.0;
  for (uint32_t t : ties) {
    const int32_t a = static_cast<int32_t>(t % static_cast<uint32_t>(c.na));
    const uint32_t xy = t / static_cast<uint32_t>(c.na);
    const int32_t xi = static_cast<int32_t>(xy % static_cast<uint32_t>(c.nx)), yi = static_cast<int32_t>(xy / static_cast<uint32_t>(c.nx));
    ax += c.center[0] + c.x_poses[xi];
    ay += c.center[1] + c.y_poses[yi];
    const double heading = normalize_angle(c.angles[a]);
    double sin_h, cos_h;
    ref_sincos(heading, &sin_h, &cos_h);
    thetaX += cos_h;
    thetaY += sin_h;
  }
  const int32_t count = static_cast<int32_t>(ties.size());
  ax /= count; ay /= count; thetaX /= count; thetaY /= count;
  const double avg[3] = {ax, ay, std::atan2(thetaY, thetaX)};

  if (!c.fine) {
    std::vector<double> lattice(plane);
    std::memcpy(lattice.data(), out + kOutHeaderWords, plane * 8);
    WalkGeometry wg;
    std::copy(c.center, c.center + 3, wg.center);
    wg.off_x = c.off_x; wg.off_y = c.off_y; wg.res_x = c.res_x; wg.res_y = c.res_y; wg.ang_res = c.ang_res;
    const int prc = positional_covariance(m, c, lattice, wg, avg, best, q.cov);
    if (prc != KH_OK) {q.status = prc; return KH_OK;}
    s.last_coarse = c; s.last_lattice.swap(lattice); s.has_last_coarse = true;     // m_pSearchSpaceProbs of this matcher slot
  } else {
    // ComputeAngularCovariance, Mapper.cpp:977-1025
    const double bestAngle = normalize_angle_difference(avg[2], c.center[2]);
    const Cell g = world_to_grid(m->scale, s.off_x, s.off_y, avg[0], avg[1]);
    const int32_t gridIndex = (g.x + m->roi_x) + (g.y + m->roi_y) * m->ws;
    // the raw responses of all angles at that cell: it is a lattice point unless the tie average
    // left the lattice, in which case the sums are recomputed by a 1x1 search at that cell
    int32_t fx = -1, fy = -1;
    for (int32_t yi = 0; yi < c.ny && fx < 0; ++yi) {
      for (int32_t xi = 0; xi < c.nx; ++xi) {
        if (c.bx[xi] + c.by[yi] == gridIndex) {fx = xi; fy = yi; break;}
      }
    }
    std::vector<int32_t> col(c.na, 0);
    if (fx >= 0 && v.small != nullptr) {
      const int32_t * vol = v.small;
      for (int32_t a = 0; a < c.na; ++a) {col[a] = vol[static_cast<size_t>(a) * plane + static_cast<size_t>(fy) * c.nx + fx];}
    } else if (!v.device_work) {
      return kNeedGeneric;
    } else if (fx >= 0) {
      KH_HIP(hipMemcpy2D(col.data(), 4, s.d_sums + static_cast<size_t>(fy) * c.nx + fx, plane * 4, 4, c.na, hipMemcpyDeviceToHost));
    } else {
      // off-lattice best pose: score the single cell through the generic (per-pose checked) path.  Finalisation may
      // run on pool threads: one of them at a time talks to the stream
      static std::mutex rescore_mutex;
      std::lock_guard<std::mutex> rescore_lock(rescore_mutex);
      s.volume_stale = true;
      CorrJob * job = v.h_job;
      CorrJob one = *job;
      one.nx = 1; one.ny = 1; one.linear = 0; one.sx = 1; one.sy_ws = m->ws; one.base0 = gridIndex;
      one.tiles_x = 1; one.tiles_y = 1; one.ry = 1; one.do_penalize = 0; one.coarse = 0; one.write_resp = 0;
      // bx/by of the single pose: reuse the first entries of the staged arrays
      int32_t one_bx = gridIndex, one_by = 0;
      KH_HIP(hipMemcpy(const_cast<int32_t *>(job->bx), &one_bx, 4, hipMemcpyHostToDevice));
      KH_HIP(hipMemcpy(const_cast<int32_t *>(job->by), &one_by, 4, hipMemcpyHostToDevice));
      KH_HIP(hipMemcpy(v.d_job, &one, sizeof(CorrJob), hipMemcpyHostToDevice));
      launch_offsets(v.d_job, v.stride, 1, one.na, m->stream);
      launch_score(v.d_job, v.stride, 1, 1, one.na, 1, 1, m->stream);
      KH_HIP(hipStreamSynchronize(m->stream));
      KH_HIP(hipMemcpy(col.data(), s.d_sums, sizeof(int32_t) * c.na, hipMemcpyDeviceToHost));
      // (the slot's stored volume now holds this 1 x 1 search: kh_matcher_read_volume reports KH_ERR_NOT_FOUND)
    }
    q.cov[8] = angular_variance(col, c.denom, c.center[2], c.ang_off, c.ang_res, bestAngle, best);
  }
  q.mean[0] = avg[0]; q.mean[1] = avg[1]; q.mean[2] = avg[2];
  q.response = best > 1.0 ? 1.0 : best;
  s.last = c; s.has_last = true;
  return KH_OK;
}

// One sub-batch of CorrelateScan jobs in two phases so that two sub-batches can be pipelined on the handle's
// stream: phase 0 = host preparation + upload + kernels + download, all enqueued, ending with an event;
// phase 1 = wait for that event + finalisation.  Everything phase 1 needs lives in the CorrBatch.
// last_of_call (phase 1 of the last chunk of a chunked call): nothing hides the host's work any more -- the worker pool is woken while
// this thread still waits for the chunk's results, and stays awake behind the finalisation for the first chunk of the caller's next call.
static int correlate_stage(kh_matcher * m, CorrReq * reqs, size_t n, CorrBatch & B, int phase, bool overlap = false, bool last_of_call = false)
{
  if (n == 0) {return KH_OK;}
  const kh_match_params & mp = m->params;
  std::vector<CorrHost> & ctx = B.ctx;
  std::vector<StageLayout> & lay = B.lay;
  size_t & stride = B.stride; size_t & out_words = B.out_words;
  int32_t & max_na = B.max_na; int32_t & max_tiles = B.max_tiles; int32_t & max_poses = B.max_poses;
  int32_t & sx_variant = B.sx_variant; int32_t & ry = B.ry;
  bool & uniform_kernel = B.uniform_kernel;
  bool & use_lds = B.use_lds;
  constexpr size_t kSmallVolume = 4096;
  int rc = KH_OK;
  // KH_MATCH_TIMING=1: wall split of the stages, printed every 64 calls (diagnostics only)
  static const bool timing = std::getenv("KH_MATCH_TIMING") != nullptr;
  static double t_acc[4] = {0, 0, 0, 0}; static long t_calls = 0;
  const auto t_enter = std::chrono::steady_clock::now();
  auto lap = [&](int slot, std::chrono::steady_clock::time_point from) {
    if (timing) {t_acc[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - from).count();}
  };
  if (phase == 0) {
  ctx.assign(n, CorrHost()); lay.assign(n, StageLayout());
  stride = 0; out_words = 0; max_na = 0; max_tiles = 0; max_poses = 0; sx_variant = -1; ry = -1; uniform_kernel = true;

  // ---- 1. host preparation (exact reference arithmetic) ----
  for (size_t i = 0; i < n; ++i) {
    CorrReq & q = reqs[i];
    CorrHost & c = ctx[i];
    q.status = KH_OK;
    rc = init_ctx(q, c);
    if (rc) {return rc;}
    lay[i] = stage_layout(c.P, c.nx, c.ny, c.na, q.penalize);
    stride = std::max(stride, lay[i].total);
    out_words = std::max(out_words, kOutHeaderWords + static_cast<size_t>(c.nx) * c.ny);
    max_na = std::max(max_na, c.na);
    max_poses = std::max(max_poses, c.nx * c.ny * c.na);
  }
  out_words = align_up(out_words, 32);
  rc = ensure_pinned(B.h_stage, B.cap_stage, stride * n, m->stream);
  if (rc) {return rc;}
  rc = ensure_device(B.d_stage, B.cap_dstage, B.cap_stage, m->stream);
  if (rc) {return rc;}
  rc = ensure_pinned(B.h_out, B.cap_hout, out_words * n, m->stream);
  if (rc) {return rc;}
  rc = ensure_device(B.d_out, B.cap_dout, out_words * n, m->stream);
  if (rc) {return rc;}

  // device scratch of every slot first (allocation is serial); the tables themselves are filled by
  // the host pool
  for (size_t i = 0; i < n; ++i) {
    rc = ensure_slot_scratch(m, reqs[i], ctx[i]);
    if (rc) {return rc;}
  }
  // LDS-staged scoring (k_offsets_lds / k_score_lds): by default for the searches it was measured faster on -- windows of at most
  // 61 bytes x 64 rows with >= 1e8 lookups per search, in launches of >= 512 angle pairs (the config-2 CorrelateScan in batches:
  // 0.48 against 0.60 ms per 51 matches); smaller searches and small launches keep the windowed kernel, whose fixed costs are lower.  KH_LDS_SCORE=1 / kh_matcher_set_debug bit 1:
  // every search the path can take; KH_LDS_SCORE=0 / bit 6: none.
  static const int lds_env = std::getenv("KH_LDS_SCORE") ? std::atoi(std::getenv("KH_LDS_SCORE")) : -1;
  const bool lds_never = lds_env == 0 || m->windowed_score;
  const bool lds_always = !lds_never && (lds_env > 0 || m->lds_score);
  std::vector<JobShape> shapes(n);
  HostPool::instance().run(n, [&](size_t i) {
    prepare_job(m, reqs[i], ctx[i], lay[i], B.h_stage + stride * i, B.d_stage + stride * i, B.d_out + out_words * i, out_words, n,
      lds_always, lds_never, shapes[i]);
  });
  if (timing) {
    const CorrJob * j0 = reinterpret_cast<const CorrJob *>(B.h_stage);
    std::fprintf(stderr, "[kh corr] job 0: %d x %d x %d poses, sx %d, tiles %d x %d of %d x %d poses, lists per tile %d, copies %d (slot kind %d), "
      "dec %d, ws %d\n", j0->nx, j0->ny, j0->na, j0->sx, j0->tiles_x, j0->tiles_y, j0->tile_px, 4 * j0->ry, j0->list_tiles,
      j0->grid2 ? 1 : 0, m->slots[ctx[0].slot].copy_kind, j0->dec, m->ws);
  }
  bool all_lds = true;
  for (size_t i = 0; i < n; ++i) {
    if (sx_variant < 0) {sx_variant = shapes[i].sx; ry = shapes[i].ry;}
    if (sx_variant != shapes[i].sx || ry != shapes[i].ry) {uniform_kernel = false;}
    max_tiles = std::max(max_tiles, shapes[i].tiles);
    all_lds = all_lds && shapes[i].lds != 0;
  }
  use_lds = all_lds && uniform_kernel;
  int32_t tile_pairs = 0;
  for (size_t i = 0; i < n; ++i) {
    CorrJob * job = reinterpret_cast<CorrJob *>(B.h_stage + stride * i);
    if (use_lds) {
      // the LDS-staged kernel scores an angle's whole lattice in one workgroup: one scoring tile per angle for K4
      job->tiles_x = 1; job->tiles_y = 1; job->ry = 16; job->tile_px = kTileSpan;
    }
    tile_pairs = std::max(tile_pairs, job->na * job->tiles_x * job->tiles_y);
  }
  B.tile_pairs = tile_pairs;

  // ---- 2. upload, launch, download ----
  lap(0, t_enter);
  const auto t_enqueue = std::chrono::steady_clock::now();
  // overlap (chunked batches): the uploads, the table / list kernel K2, the tie scan K4 and the downloads go on the
  // side stream, ordered against the scoring kernel by events: the main stream then runs K3 after K3, and everything
  // else of a chunk happens under the scoring of its neighbours
  hipStream_t cs = overlap ? B.side : m->stream;
  KH_HIP(hipMemcpyAsync(B.d_stage, B.h_stage, stride * n, hipMemcpyHostToDevice, cs));
  // (the result blocks are zeroed by K2)
  if (m->profiling) {KH_HIP(hipEventRecord(B.evs[0], cs));}
  if (use_lds) {
    launch_offsets_lds(B.d_stage, stride, static_cast<int32_t>(n), max_na, cs);
  } else {
    launch_offsets(B.d_stage, stride, static_cast<int32_t>(n), max_na, cs);
  }
  if (m->profiling) {KH_HIP(hipEventRecord(B.evs[1], cs));}
  // The scoring kernel of a chunk goes on the chunk's own side stream as well (KH_K3_MAIN=1: on the handle's main stream, one
  // scoring kernel after the other, as through round 3): the two staging sets then are two independent in-order queues, and the
  // workgroups of chunk i + 1 fill the compute units the tail of chunk i leaves idle (2099 workgroups on 512 slots are 4.1
  // rounds: a fifth of the kernel's time ran at a tenth of the occupancy).
  static const bool k3_main = std::getenv("KH_K3_MAIN") != nullptr;
  hipStream_t ks = (overlap && !k3_main) ? cs : m->stream;
  if (overlap && ks != cs) {
    KH_HIP(hipEventRecord(B.up, cs));
    KH_HIP(hipStreamWaitEvent(ks, B.up, 0));
  }
  if (m->profiling) {KH_HIP(hipEventRecord(B.ev[0], ks));}
  if (use_lds) {
    bool full_rows = true;
    for (size_t i = 0; i < n; ++i) {full_rows = full_rows && lds_row_waves(ctx[i].ny) == 4;}
    launch_score_lds(B.d_stage, stride, static_cast<int32_t>(n), max_na, sx_variant, full_rows, ks);
  } else if (uniform_kernel) {
    launch_score(B.d_stage, stride, static_cast<int32_t>(n), max_tiles, max_na, sx_variant, ry, ks, m->mfma_score);
  } else {
    for (size_t i = 0; i < n; ++i) {
      const CorrJob * job = reinterpret_cast<const CorrJob *>(B.h_stage + stride * i);
      launch_score(B.d_stage + stride * i, stride, 1, job->tiles_x * job->tiles_y, job->na,
        (job->linear && job->sx == 2 && !job->dec) ? 2 : 1, job->ry, ks, m->mfma_score);
    }
  }
  if (m->profiling) {KH_HIP(hipEventRecord(B.ev[1], ks));}
  if (overlap && ks != cs) {
    KH_HIP(hipEventRecord(B.kdone, ks));
    KH_HIP(hipStreamWaitEvent(cs, B.kdone, 0));
  }
  hipStream_t ts = cs;
  if (m->profiling) {KH_HIP(hipEventRecord(B.evs[2], ts));}
  launch_ties(B.d_stage, stride, static_cast<int32_t>(n), max_poses, B.tile_pairs, ts);
  if (m->profiling) {KH_HIP(hipEventRecord(B.evs[3], ts));}
  KH_HIP(hipGetLastError());
  KH_HIP(hipMemcpyAsync(B.h_out, B.d_out, out_words * 8 * n, hipMemcpyDeviceToHost, ts));
  // fine passes need the raw sums of every angle at the best cell (ComputeAngularCovariance): their
  // volumes are tiny (3 x 3 x nA), so they ride along with the batch download instead of costing one
  // synchronous copy per match afterwards
  {
    size_t max_small = 0;
    for (size_t i = 0; i < n; ++i) {
      const size_t vol = static_cast<size_t>(ctx[i].nx) * ctx[i].ny * ctx[i].na;
      if (ctx[i].fine && vol <= kSmallVolume) {max_small = std::max(max_small, vol);}
    }
    B.small_stride = align_up(max_small, 32);
    if (max_small) {
      rc = ensure_pinned(B.h_sums, B.cap_hsums, B.small_stride * n, m->stream); if (rc) {return rc;}
      rc = ensure_device(B.d_small, B.cap_dsmall, B.small_stride * n, m->stream); if (rc) {return rc;}
      launch_gather_small(B.d_stage, stride, static_cast<int32_t>(n), B.d_small, static_cast<int32_t>(B.small_stride), ts);
      KH_HIP(hipMemcpyAsync(B.h_sums, B.d_small, B.small_stride * n * 4, hipMemcpyDeviceToHost, ts));
    }
  }
  KH_HIP(hipEventRecord(B.done, ts));
  lap(1, t_enqueue);
  return KH_OK;
  }   // phase 0
  // (what a wake-up costs -- a futex wake and the scheduler, ~50 us -- is hidden under the scoring kernels for every chunk but the
  // last, and for every preparation but the first of the next call)
  constexpr uint64_t kAwaitResults = 200 * HostPool::kTicksPerMicrosecond, kAwaitNextCall = 120 * HostPool::kTicksPerMicrosecond;
  if (last_of_call) {HostPool::instance().run(2, [](size_t) {}, kAwaitResults);}
  KH_HIP(hipEventSynchronize(B.done));
  lap(2, t_enter);
  const auto t_final = std::chrono::steady_clock::now();
  if (use_lds && std::getenv("KH_LDS_DEBUG")) {
    const CorrHost & c0 = ctx[0];
    const Slot & s0 = m->slots[c0.slot];
    const size_t groups = (static_cast<size_t>(c0.na) + kGroupAngles - 1) / kGroupAngles;
    const size_t range_len = static_cast<size_t>(lds_desc_capacity(c0.P));
    std::vector<int32_t> cc(groups * kLdsRanges), dd(groups * kLdsRanges * range_len * kChunkWords);
    KH_HIP(hipMemcpy(cc.data(), s0.d_chunk_counts, cc.size() * 4, hipMemcpyDeviceToHost));
    KH_HIP(hipMemcpy(dd.data(), s0.d_chunks, dd.size() * 4, hipMemcpyDeviceToHost));
    long total = 0, bytes = 0, windows = 0, maxb = 0, small = 0;
    for (size_t g = 0; g < groups * kLdsRanges; ++g) {
      for (int32_t k = 0; k < cc[g]; ++k) {
        const int32_t * d = dd.data() + (g * range_len + k) * kChunkWords;
        ++total; bytes += static_cast<long>(d[2]) * kLdsPitch; windows += d[3];
        maxb = std::max<long>(maxb, static_cast<long>(d[2]) * kLdsPitch);
        if (d[3] < 16) {++small;}
      }
    }
    std::fprintf(stderr, "[kh lds] job 0: %zu angle pairs, %ld chunks (%.1f per pair, %ld with < 16 windows), mean region %.1f KB, max %.1f KB, "
      "%.1f windows per chunk, %.0f staged bytes per window\n",
      groups, total, static_cast<double>(total) / groups, small, bytes / 1024.0 / std::max(1l, total), maxb / 1024.0,
      static_cast<double>(windows) / std::max(1l, total), static_cast<double>(bytes) / std::max(1l, windows));
  }
  if (m->profiling) {
    float ms = 0;
    KH_HIP(hipEventElapsedTime(&ms, B.ev[0], B.ev[1]));
    m->score_ms += ms; m->score_launches += 1; m->score_jobs += static_cast<int64_t>(n);
    KH_HIP(hipEventElapsedTime(&ms, B.evs[0], B.evs[1])); m->offsets_ms += ms;
    KH_HIP(hipEventElapsedTime(&ms, B.evs[2], B.evs[3])); m->ties_ms += ms;
  }

  // ---- 3. finalisation (Mapper.cpp:775-862) ----
  std::vector<int> final_rc(n, KH_OK);
  auto finalize = [&](size_t i) -> int {
    ResultView v;
    v.out = B.h_out + out_words * i;
    const size_t vol = static_cast<size_t>(ctx[i].nx) * ctx[i].ny * ctx[i].na;
    v.small = (ctx[i].fine && vol <= kSmallVolume && B.h_sums) ? B.h_sums + B.small_stride * i : nullptr;
    v.h_job = reinterpret_cast<CorrJob *>(B.h_stage + stride * i); v.d_job = B.d_stage + stride * i; v.stride = stride;
    return finalize_job(m, reqs[i], ctx[i], v);
  };
  // big fine volumes and the off-lattice re-score path issue their own copies / launches on the
  // stream: keep such batches off the pool
  bool serial_final = false;
  for (size_t i = 0; i < n; ++i) {
    serial_final = serial_final || (ctx[i].fine && static_cast<size_t>(ctx[i].nx) * ctx[i].ny * ctx[i].na > kSmallVolume);
  }
  if (serial_final) {
    for (size_t i = 0; i < n; ++i) {final_rc[i] = finalize(i);}
  } else {
    HostPool::instance().run(n, [&](size_t i) {
      (void)hipSetDevice(m->device);
      final_rc[i] = finalize(i);
    }, last_of_call ? kAwaitNextCall : 0);
  }
  lap(3, t_final);
  static const long period = (std::getenv("KH_MATCH_TIMING") && std::atoi(std::getenv("KH_MATCH_TIMING")) > 1) ? 1 : 64;
  if (timing && ++t_calls % period == 0) {
    std::fprintf(stderr, "[kh match] per call: prepare %.3f ms, enqueue %.3f ms, wait %.3f ms, finalize %.3f ms (n = %zu, stage %zu B/job)\n",
      t_acc[0] / period, t_acc[1] / period, t_acc[2] / period, t_acc[3] / period, n, stride);
    t_acc[0] = t_acc[1] = t_acc[2] = t_acc[3] = 0;
  }
  for (size_t i = 0; i < n; ++i) {if (final_rc[i] != KH_OK) {return final_rc[i];}}
  return KH_OK;
}

int correlate_batch(kh_matcher * m, std::vector<CorrReq> & reqs)
{
  const size_t n = reqs.size();
  if (n == 0) {return KH_OK;}
  // Large batches go through in chunks of kChunk jobs on the two staging sets of the handle: while the kernels of
  // chunk i run, the host prepares chunk i + 1 and finalises chunk i - 1 (the exact host half costs ~5 us per match
  // on the worker pool, the scoring kernel ~13 us).  Chunks of 64 keep the scoring launches at full efficiency;
  // splitting a batch of 32 into halves was measured slower than not splitting (0.85 against 0.76 ms), so batches
  // below 2 * kChunk are not split.  KH_PIPELINE=0 switches the chunking off.
  static const size_t kChunk = std::getenv("KH_CHUNK") ? static_cast<size_t>(std::max(8, std::atoi(std::getenv("KH_CHUNK")))) : 64;
  static const bool pipeline = !(std::getenv("KH_PIPELINE") && std::atoi(std::getenv("KH_PIPELINE")) == 0);
  // ... and only searches whose scoring kernel dwarfs the hand-overs between the streams: a chunk of 64 config-2
  // searches scores for 0.85 ms, a chunk of loop-closure coarse searches (half the lookups, a quarter of the loads)
  // for 0.2 ms, and batches of those were measured 25 % slower chunked than whole
  const CorrReq & r0 = reqs[0];
  const double work0 = (round_half_away(r0.off_x * 2.0 / r0.res_x) + 1) * (round_half_away(r0.off_y * 2.0 / r0.res_y) + 1) *
    (round_half_away(r0.ang_off * 2.0 / r0.ang_res) + 1) * static_cast<double>(r0.scan->n);
  if (!pipeline || n < 2 * kChunk || (work0 < 2.5e8 && !m->force_chunks)) {
    int rc = correlate_stage(m, reqs.data(), n, m->batch[0], 0);
    if (rc) {return rc;}
    return correlate_stage(m, reqs.data(), n, m->batch[0], 1);
  }
  // the side streams start behind everything already queued on the main stream (the rasteriser writes the grids
  // and the occupancy block maps K2 reads); they are drained before this call returns (every chunk's `done` is
  // waited for), so later main-stream work needs no edge back
  KH_HIP(hipEventRecord(m->batch[0].kdone, m->stream));
  for (auto & b : m->batch) {KH_HIP(hipStreamWaitEvent(b.side, m->batch[0].kdone, 0));}
  // a shorter first chunk: the pipeline fills on it (its preparation, upload and K2 are exposed) and drains on the last (its
  // K4, download and finalisation are).  256 config-2 matches as 24 + 64 + 64 + 64 + 40 measured 79.9 k and 81.4 k matches/s
  // on two boxes where 32 + 64 + 64 + 64 + 32 (the split of rounds 2-3) gave 78.4 k and 76.5 k, a first chunk of 16 74.4 k,
  // uniform chunks of 64 less again; splits sized to whole rounds of workgroups (24 + 62 + 62 + 62 + 46) measured no better
  // than the plain one.  KH_CHUNK_EDGE=64 restores the uniform split, KH_CHUNK_SIZES=a,b,... sets one by hand.
  static const size_t edge = std::getenv("KH_CHUNK_EDGE") ? static_cast<size_t>(std::max(8, std::min(256, std::atoi(std::getenv("KH_CHUNK_EDGE"))))) : 24;
  std::vector<size_t> bounds;
  // KH_CHUNK_SIZES=a,b,c,...: an explicit split (measurements); the last size is repeated / cut to cover the batch
  static const char * sizes_env = std::getenv("KH_CHUNK_SIZES");
  if (sizes_env) {
    size_t at = 0, last = kChunk;
    const char * p = sizes_env;
    while (at < n) {
      if (*p) {
        char * end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end != p && v > 0) {last = static_cast<size_t>(v);}
        p = (*end == ',') ? end + 1 : end;
      }
      bounds.push_back(at);
      at += last;
    }
  } else if (edge < kChunk && n >= 2 * edge + kChunk) {
    bounds.push_back(0);
    for (size_t at = edge; at + edge < n; at += kChunk) {bounds.push_back(at);}
    if (n - bounds.back() > kChunk) {bounds.push_back(n - edge);}
  } else {
    for (size_t at = 0; at < n; at += kChunk) {bounds.push_back(at);}
  }
  bounds.push_back(n);
  const size_t chunks = bounds.size() - 1;
  auto begin_of = [&](size_t c) {return bounds[c];};
  auto size_of = [&](size_t c) {return bounds[c + 1] - bounds[c];};
  int first_rc = KH_OK;
  int rc = correlate_stage(m, reqs.data(), size_of(0), m->batch[0], 0, true);
  if (rc) {return rc;}
  for (size_t c = 1; c < chunks; ++c) {
    rc = correlate_stage(m, reqs.data() + begin_of(c), size_of(c), m->batch[c & 1], 0, true);
    if (rc) {(void)hipStreamSynchronize(m->stream); for (auto & b : m->batch) {(void)hipStreamSynchronize(b.side);} return rc;}
    rc = correlate_stage(m, reqs.data() + begin_of(c - 1), size_of(c - 1), m->batch[(c - 1) & 1], 1, true);
    if (rc && !first_rc) {first_rc = rc;}
  }
  rc = correlate_stage(m, reqs.data() + begin_of(chunks - 1), size_of(chunks - 1), m->batch[(chunks - 1) & 1], 1, true, true);
  return first_rc ? first_rc : rc;
}

}  // namespace kh

namespace kh
{
// for the other translation units of the library (kh_matcher is private to this one)
int32_t matcher_max_batch(const kh_matcher * m) {return m ? m->max_batch : 0;}
}

// =============================================================================================
//                                         C ABI
// =============================================================================================
extern "C" {

const char * kh_last_error(void) {return kh::g_last_error.c_str();}

int kh_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {return 0;}
  return n;
}

const char * kh_version(void) {return "karto-hip 0.1 (gfx950)";}

int kh_scan_points(const double * ranges, int32_t n, const double sensor_pose[3], double min_angle,
  double angular_resolution, double * out)
{
  if (!ranges || !out || n < 0) {return KH_ERR_INVALID_ARG;}
  for (int32_t i = 0; i < n; ++i) {          // Karto.h:5663-5682
    const double angle = sensor_pose[2] + min_angle + static_cast<uint32_t>(i) * angular_resolution;
    double sin_a, cos_a;
    ref_sincos(angle, &sin_a, &cos_a);
    out[2 * i] = sensor_pose[0] + (ranges[i] * cos_a);
    out[2 * i + 1] = sensor_pose[1] + (ranges[i] * sin_a);
  }
  return KH_OK;
}

void kh_match_params_default(kh_match_params * p)
{
  // Mapper.cpp:2250-2293
  p->coarse_search_angle_offset = 20 * kPi180;
  p->coarse_angle_resolution = 2 * kPi180;
  p->fine_search_angle_offset = 0.2 * kPi180;
  p->use_response_expansion = 0;
  p->distance_variance_penalty = 0.3 * 0.3;
  p->minimum_distance_penalty = 0.5;
  p->angle_variance_penalty = (20 * kPi180) * (20 * kPi180);
  p->minimum_angle_penalty = 0.9;
}

int kh_matcher_create(double search_size, double resolution, double smear, double range_threshold,
  int32_t device, int32_t max_batch, kh_matcher ** out)
{
  if (!out) {return KH_ERR_INVALID_ARG;}
  *out = nullptr;
  // Mapper.cpp:481-493
  if (resolution <= 0 || search_size <= 0 || smear < 0 || range_threshold <= 0 || max_batch < 1) {
    set_error("ScanMatcher::Create: invalid parameters");
    return KH_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    set_error("no usable HIP device (libkartohip has no CPU fallback)");
    return KH_ERR_NO_DEVICE;
  }
  kh_matcher * m = new kh_matcher();
  m->search_size = search_size; m->resolution = resolution; m->smear = smear; m->range_threshold = range_threshold;
  m->device = device; m->max_batch = max_batch;
  kh_match_params_default(&m->params);
  // Mapper.cpp:498-505
  const uint32_t side = static_cast<uint32_t>(round_half_away(search_size / resolution) + 1);
  const uint32_t margin = static_cast<uint32_t>(std::ceil(range_threshold / resolution));
  const int32_t grid_size = static_cast<int32_t>(side + 2 * margin);
  // CorrelationGrid::CreateGrid / ctor, Mapper.h:1099-1114, 1194-1208; Grid::Resize Karto.h:4636-4664
  const uint32_t border = static_cast<uint32_t>(half_kernel_size(smear, resolution)) + 1;
  m->width = grid_size + 2 * static_cast<int32_t>(border);
  m->height = m->width;
  m->ws = static_cast<int32_t>((static_cast<size_t>(m->width) + 7) & ~static_cast<size_t>(7));
  const int64_t ds = static_cast<int64_t>(m->ws) * m->height;
  if (ds <= 0 || ds > (1ll << 31) - 4096) {delete m; set_error("grid too large"); return KH_ERR_INVALID_ARG;}
  m->data_size = static_cast<int32_t>(ds);
  m->scale = 1.0 / resolution;
  m->roi_x = m->roi_y = static_cast<int32_t>(border);
  m->roi_w = m->roi_h = grid_size;
  m->side = static_cast<int32_t>(side);
  // CalculateKernel, Mapper.h:1213-1266
  const double res = m->grid_resolution();
  const double min_dev = 0.5 * res, max_dev = 10 * res;
  if (!(smear >= min_dev && smear <= max_dev)) {
    delete m;
    set_error("Mapper Error:  Smear deviation too small / too large");
    return KH_ERR_INVALID_ARG;
  }
  m->kernel_size = 2 * half_kernel_size(smear, res) + 1;
  const int32_t k = m->kernel_size, hk = k / 2;
  m->kernel.resize(static_cast<size_t>(k) * k);
  for (int32_t i = -hk; i <= hk; ++i) {
    for (int32_t j = -hk; j <= hk; ++j) {
      const double distance_from_mean = hypot(i * res, j * res);
      const double z = exp(-0.5 * pow(distance_from_mean / smear, 2));
      const uint32_t v = static_cast<uint32_t>(round_half_away(z * 100));
      m->kernel[(i + hk) + k * (j + hk)] = static_cast<uint8_t>(v);
      if (v == 100) {m->footprint100.push_back(Cell{i, j});}
    }
  }

  if (m->footprint100.size() > static_cast<size_t>(kMaxFootprint) + 1) {
    // cannot happen inside [0.5, 10] x resolution: the diagonal neighbours reach 99 at sigma / res = 10
    delete m; set_error("smear kernel with more than five cells of 100"); return KH_ERR_INVALID_ARG;
  }
  auto fail = [&](hipError_t e, const char * what) {
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    kh_matcher_destroy(m);
    return KH_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) {return fail(e, "hipSetDevice");}
  if ((e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking)) != hipSuccess) {return fail(e, "hipStreamCreate");}
  for (auto & ev : m->ev) {
    if ((e = hipEventCreate(&ev)) != hipSuccess) {return fail(e, "hipEventCreate");}
  }
  for (auto & b : m->batch) {
    for (auto & ev : b.ev) {
      if ((e = hipEventCreate(&ev)) != hipSuccess) {return fail(e, "hipEventCreate");}
    }
    for (auto & ev : b.evs) {
      if ((e = hipEventCreate(&ev)) != hipSuccess) {return fail(e, "hipEventCreate");}
    }
    if ((e = hipEventCreateWithFlags(&b.done, hipEventDisableTiming)) != hipSuccess) {return fail(e, "hipEventCreate");}
    if ((e = hipEventCreateWithFlags(&b.up, hipEventDisableTiming)) != hipSuccess) {return fail(e, "hipEventCreate");}
    if ((e = hipEventCreateWithFlags(&b.kdone, hipEventDisableTiming)) != hipSuccess) {return fail(e, "hipEventCreate");}
    if ((e = hipStreamCreateWithFlags(&b.side, hipStreamNonBlocking)) != hipSuccess) {return fail(e, "hipStreamCreate");}
  }
  if ((e = hipMalloc(reinterpret_cast<void **>(&m->d_kernel), m->kernel.size())) != hipSuccess) {return fail(e, "hipMalloc kernel");}
  // (word 0 = the tally; the words behind it are used by measurement builds of the kernels only)
  if ((e = hipMalloc(reinterpret_cast<void **>(&m->d_load_counter), 512)) != hipSuccess) {return fail(e, "hipMalloc counter");}
  if ((e = hipMemset(m->d_load_counter, 0, 512)) != hipSuccess) {return fail(e, "hipMemset counter");}
  if ((e = hipMemcpy(m->d_kernel, m->kernel.data(), m->kernel.size(), hipMemcpyHostToDevice)) != hipSuccess) {return fail(e, "hipMemcpy kernel");}
  // zero rows either side: as many as the SEARCH SPACE is high (m->side cells: 61 / 161 / 51 for the C2 / L / S presets -- not the
  // grid's region of interest, which also spans the range threshold), so that any search window that touches the array is inside
  m->pad_rows = m->side + 8;
  m->grid_pad = align_up(static_cast<size_t>(m->pad_rows) * m->ws, 256) + kGridPad;
  m->pitch2 = static_cast<int32_t>(align_up(static_cast<size_t>(m->ws), 128));
  {
    const size_t copy = align_up(static_cast<size_t>(m->pitch2) * (m->data_size / m->ws + 2 * m->pad_rows) + 256, 256);
    if (2 * copy + 256 > (1ull << 31) - 4096) {m->dual_copy = false;}        // offsets into the copies are int32
    m->copy_b = static_cast<int32_t>(std::min<size_t>(copy, (1ull << 30)));
  }
  {
    m->pitch_d = static_cast<int32_t>(align_up(static_cast<size_t>(m->ws) / 2 + 64, 128));
    const size_t quarter = align_up(static_cast<size_t>(m->pitch_d) * (m->data_size / m->ws + 2 * m->pad_rows) + 256, 256);
    m->copy_q = 4 * quarter + 4096 < (1ull << 31) ? static_cast<int32_t>(quarter) : 0;
  }
  m->rt_w = (m->ws + kRasterTile - 1) / kRasterTile;
  m->rt_h = (m->data_size / m->ws + kRasterTile - 1) / kRasterTile;
  // occupancy block map: 8 x 8-cell blocks where the searches are small, a stamp is, and the handle is made for batches (a stamp marks every block its footprint
  // overlaps: 41 x 41 cells are 49 blocks of 8 x 8), 32 x 32-cell blocks otherwise.  On the config-2 search the finer map leaves
  // out 9 % more windows.
  // (handles made for batches: one match at a time builds its map inside one workgroup, where the coarse map is the cheaper one)
  m->bshift = (m->kernel_size <= 25 && m->side <= 64 && max_batch >= 8) ? 3 : kBlockShift;
  m->bm_w = (((m->ws >> m->bshift) + 1) + 31) / 32 + 1;     // words per block row (+1 padding word)
  m->bm_h = (m->data_size / m->ws >> m->bshift) + 2;
  // Batches take the first-point rasteriser (matcher_seq.hip) where every slot can have its table over the region of interest --
  // 4 bytes per cell: 66 MB for the sequential preset, 260 MB for the config-2 geometry -- within 24 GB per handle; KH_TABLE_RASTER=0
  // keeps the hash-table passes (measurements)
  {
    const double table_bytes = 4.0 * static_cast<double>(m->roi_w) * m->roi_h * max_batch;
    static const bool env_off = std::getenv("KH_TABLE_RASTER") != nullptr && std::atoi(std::getenv("KH_TABLE_RASTER")) == 0;
    m->table_raster = !env_off && table_bytes <= 24e9 && m->rt_w * m->rt_h <= 16384;
  }
  m->slots.resize(max_batch);
  for (auto & s : m->slots) {
    if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_grid_alloc), static_cast<size_t>(m->data_size) + 2 * m->grid_pad)) != hipSuccess) {return fail(e, "hipMalloc grid");}
    if ((e = hipMemset(s.d_grid_alloc, 0, static_cast<size_t>(m->data_size) + 2 * m->grid_pad)) != hipSuccess) {return fail(e, "hipMemset grid");}
    s.d_grid = s.d_grid_alloc + m->grid_pad;
    if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_blockmap), static_cast<size_t>(m->bm_w) * m->bm_h * 4)) != hipSuccess) {return fail(e, "hipMalloc block map");}
    if ((e = hipMemset(s.d_blockmap, 0, static_cast<size_t>(m->bm_w) * m->bm_h * 4)) != hipSuccess) {return fail(e, "hipMemset block map");}
    if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_rtiles), (4 * static_cast<size_t>(m->rt_w) * m->rt_h + 8) * sizeof(int32_t))) != hipSuccess) {return fail(e, "hipMalloc raster tiles");}
    // tiles the slot's previous rasterisation wrote ([0] = how many): none yet, the grid is all zero
    const size_t prev_bytes = (static_cast<size_t>(m->rt_w) * m->rt_h + 4) * sizeof(int32_t);
    if ((e = hipMalloc(reinterpret_cast<void **>(&s.d_prev_work), prev_bytes)) != hipSuccess) {return fail(e, "hipMalloc tile list");}
    if ((e = hipMemset(s.d_prev_work, 0, prev_bytes)) != hipSuccess) {return fail(e, "hipMemset tile list");}
  }
  if ((e = hipHostMalloc(reinterpret_cast<void **>(&m->h_rjobs), sizeof(RasterJob) * max_batch, hipHostMallocDefault)) != hipSuccess) {return fail(e, "hipHostMalloc");}
  if ((e = hipMalloc(reinterpret_cast<void **>(&m->d_rjobs), sizeof(RasterJob) * max_batch)) != hipSuccess) {return fail(e, "hipMalloc");}
  *out = m;
  return KH_OK;
}

void kh_matcher_destroy(kh_matcher * m)
{
  if (!m) {return;}
  hipSetDevice(m->device);
  if (m->stream) {hipStreamSynchronize(m->stream);}
  seq_destroy(m);
  for (auto & s : m->slots) {
    hipFree(s.d_grid_alloc); hipFree(s.d_grid2_alloc); hipFree(s.d_prev_work); hipFree(s.d_blockmap); hipFree(s.d_rtiles); hipFree(s.d_rlists); hipFree(s.d_tile_best); hipFree(s.d_table); hipFree(s.d_fast); hipFree(s.d_tcounts); hipFree(s.d_slow); hipFree(s.d_counts);
    hipFree(s.d_chunks); hipFree(s.d_chunk_counts);
    hipFree(s.d_sums); hipFree(s.d_resp); hipFree(s.d_ractive);
    hipFree(s.d_hkeys); hipFree(s.d_hvals); hipFree(s.d_hstate); hipFree(s.d_hnbr);
    hipFree(s.d_first); hipFree(s.d_cand); hipFree(s.d_seqctl); hipFree(s.d_work2);
  }
  hipFree(m->d_kernel); hipFree(m->d_rjobs); hipFree(m->d_load_counter); hipFree(m->d_tab);
  for (auto & b : m->batch) {
    hipFree(b.d_stage); hipFree(b.d_out); hipFree(b.d_small);
    if (b.h_stage) {hipHostFree(b.h_stage);}
    if (b.h_out) {hipHostFree(b.h_out);}
    if (b.h_sums) {hipHostFree(b.h_sums);}
    for (auto & ev : b.ev) {if (ev) {hipEventDestroy(ev);}}
    for (auto & ev : b.evs) {if (ev) {hipEventDestroy(ev);}}
    if (b.done) {hipEventDestroy(b.done);}
    if (b.up) {hipEventDestroy(b.up);}
    if (b.kdone) {hipEventDestroy(b.kdone);}
    if (b.side) {hipStreamSynchronize(b.side); hipStreamDestroy(b.side);}
  }
  hipFree(m->d_arena); hipFree(m->d_meta);
  if (m->h_arena) {hipHostFree(m->h_arena);}
  if (m->h_meta) {hipHostFree(m->h_meta);}
  if (m->h_rjobs) {hipHostFree(m->h_rjobs);}
  for (auto & ev : m->ev) {if (ev) {hipEventDestroy(ev);}}
  if (m->stream) {hipStreamDestroy(m->stream);}
  delete m;
}

int kh_matcher_set_params(kh_matcher * m, const kh_match_params * p)
{
  if (!m || !p) {return KH_ERR_INVALID_ARG;}
  if (!(p->coarse_angle_resolution != 0.0) || !(p->fine_search_angle_offset != 0.0) ||
    p->minimum_distance_penalty < 0.0 || p->minimum_angle_penalty < 0.0)
  {
    set_error("invalid match parameters");
    return KH_ERR_INVALID_ARG;
  }
  m->params = *p;
  return KH_OK;
}

int kh_matcher_set_debug(kh_matcher * m, int32_t keep_response_volume)
{
  if (!m) {return KH_ERR_INVALID_ARG;}
  m->keep_responses = (keep_response_volume & 1) != 0;
  m->lds_score = (keep_response_volume & 2) != 0;
  m->windowed_score = (keep_response_volume & 64) != 0;
  m->no_seq = (keep_response_volume & 128) != 0;
  m->dense_score = (keep_response_volume & 4) != 0;
  if (keep_response_volume & 32) {m->mfma_score = true;}
  m->force_chunks = (keep_response_volume & 8) != 0 || std::getenv("KH_FORCE_CHUNKS") != nullptr;
  m->dual_copy = (keep_response_volume & 16) == 0;
  return KH_OK;
}

static int check_scan(const kh_scan * s)
{
  if (!s || s->n < 0 || (s->n > 0 && (!s->ranges || !s->points_xy))) {return KH_ERR_INVALID_ARG;}
  return KH_OK;
}

int kh_matcher_add_scans(kh_matcher * m, int32_t slot, const kh_scan * query, const kh_scan * base, int32_t n_base)
{
  if (!m || slot < 0 || slot >= m->max_batch || !query || n_base < 0 || (n_base > 0 && !base)) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  std::vector<RasterReq> reqs{RasterReq{slot, query, base, n_base}};
  int rc = raster_batch(m, reqs);
  if (rc) {return rc;}
  KH_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

int kh_matcher_correlate_batch(kh_matcher * m, int32_t n, const kh_scan * queries, const double * centers,
  const double search_offset[2], const double search_resolution[2], double angle_offset,
  double angle_resolution, int32_t do_penalize, int32_t fine, double * means, double * covs,
  double * responses, int32_t * status)
{
  if (!m || n < 0 || n > m->max_batch || !queries || !centers || !means || !covs || !responses) {return KH_ERR_INVALID_ARG;}
  if (!(search_resolution[0] > 0.0) || !(search_resolution[1] > 0.0) || angle_resolution == 0.0) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  std::vector<CorrReq> reqs(n);
  for (int32_t i = 0; i < n; ++i) {
    if (check_scan(&queries[i]) != KH_OK || queries[i].n == 0) {return KH_ERR_INVALID_ARG;}
    CorrReq & q = reqs[i];
    q.slot = i; q.scan = &queries[i];
    std::copy(centers + 3 * i, centers + 3 * i + 3, q.center);
    q.off_x = search_offset[0]; q.off_y = search_offset[1];
    q.res_x = search_resolution[0]; q.res_y = search_resolution[1];
    q.ang_off = angle_offset; q.ang_res = angle_resolution;
    q.penalize = do_penalize != 0; q.fine = fine != 0;
    std::copy(covs + 9 * i, covs + 9 * i + 9, q.cov);
    q.response = 0; q.status = KH_OK;
  }
  int rc = correlate_batch(m, reqs);
  if (rc) {return rc;}
  int worst = KH_OK;
  for (int32_t i = 0; i < n; ++i) {
    std::copy(reqs[i].mean, reqs[i].mean + 3, means + 3 * i);
    std::copy(reqs[i].cov, reqs[i].cov + 9, covs + 9 * i);
    responses[i] = reqs[i].response;
    if (status) {status[i] = reqs[i].status;}
    if (reqs[i].status != KH_OK) {worst = reqs[i].status;}
  }
  return status ? KH_OK : worst;
}

int kh_matcher_correlate(kh_matcher * m, int32_t slot, const kh_scan * query, const double center[3],
  const double search_offset[2], const double search_resolution[2], double angle_offset,
  double angle_resolution, int32_t do_penalize, int32_t fine, double mean[3], double cov[9], double * response)
{
  if (!m || slot < 0 || slot >= m->max_batch || check_scan(query) != KH_OK || query->n == 0 || !mean || !cov || !response) {
    return KH_ERR_INVALID_ARG;
  }
  if (!(search_resolution[0] > 0.0) || !(search_resolution[1] > 0.0) || angle_resolution == 0.0) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  std::vector<CorrReq> reqs(1);
  CorrReq & q = reqs[0];
  q.slot = slot; q.scan = query;
  std::copy(center, center + 3, q.center);
  q.off_x = search_offset[0]; q.off_y = search_offset[1]; q.res_x = search_resolution[0]; q.res_y = search_resolution[1];
  q.ang_off = angle_offset; q.ang_res = angle_resolution; q.penalize = do_penalize != 0; q.fine = fine != 0;
  std::copy(cov, cov + 9, q.cov);
  q.response = 0; q.status = KH_OK;
  int rc = correlate_batch(m, reqs);
  if (rc) {return rc;}
  if (q.status != KH_OK) {return q.status;}
  std::copy(q.mean, q.mean + 3, mean);
  std::copy(q.cov, q.cov + 9, cov);
  *response = q.response;
  return KH_OK;
}

// MatchScan for n independent (query, base chain) pairs, stage by stage over the whole batch
int kh_matcher_match_batch(kh_matcher * m, int32_t n, const kh_scan * queries, const kh_scan * base,
  const int32_t * base_begin, int32_t do_penalize, int32_t do_refine, double * means, double * covs,
  double * responses, int32_t * status)
{
  if (!m || n < 0 || n > m->max_batch || !queries || !base_begin || !means || !covs || !responses) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  static const bool batch_timing = std::getenv("KH_MATCH_TIMING") != nullptr;
  const auto t_batch = std::chrono::steady_clock::now();
  struct Report {
    bool on; std::chrono::steady_clock::time_point t0; int32_t n;
    ~Report() {
      if (on) {std::fprintf(stderr, "[kh match_batch] %d matches, %.3f ms inside the call\n", n,
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());}
    }
  } report{batch_timing, t_batch, n};
  const kh_match_params & mp = m->params;
  // (a query whose readings the caller finishes later -- QueryHook, matcher_private.hpp: whatever way this call takes, they are
  // complete before it reads them, and when it returns)
  struct HookGuard {~HookGuard() {pending_query_hook().run();}} hook_guard;
  std::vector<int> st(n, KH_OK);
  std::vector<int32_t> active;
  std::vector<RasterReq> rreqs;
  for (int32_t i = 0; i < n; ++i) {
    if (check_scan(&queries[i]) != KH_OK) {return KH_ERR_INVALID_ARG;}
    const int32_t nb = base_begin[i + 1] - base_begin[i];
    if (nb < 0 || (nb > 0 && !base)) {return KH_ERR_INVALID_ARG;}
    for (int32_t b = 0; b < nb; ++b) {
      if (check_scan(&base[base_begin[i] + b]) != KH_OK) {return KH_ERR_INVALID_ARG;}
    }
    double * mean = means + 3 * i; double * cov = covs + 9 * i;
    std::fill(cov, cov + 9, 0.0);
    if (queries[i].n == 0) {
      // Mapper.cpp:547-557
      std::copy(queries[i].sensor_pose, queries[i].sensor_pose + 3, mean);
      cov[0] = kMaxVariance; cov[4] = kMaxVariance;
      cov[8] = 4 * (mp.coarse_angle_resolution * mp.coarse_angle_resolution);
      responses[i] = 0.0;
      continue;
    }
    active.push_back(i);
    rreqs.push_back(RasterReq{i, &queries[i], base ? base + base_begin[i] : nullptr, nb});
  }
  // ONE MatchScan -- what the reference's API issues (Mapper.cpp:2714-2717) -- takes the fused path (matcher_seq.cpp); it hands
  // back, pass by pass, whatever it could not finish
  int rc = KH_OK;
  bool coarse_done = false, fine_done = false;
  if (n == 1 && active.size() == 1) {
    int seq_status = KH_OK;
    rc = seq_match(m, &queries[0], base ? base + base_begin[0] : nullptr, base_begin[1] - base_begin[0], do_penalize != 0, do_refine != 0,
      means, covs, responses, &seq_status, &coarse_done, &fine_done);
    if (rc) {return rc;}
    if (coarse_done) {st[0] = seq_status;}
  }
  if (!coarse_done) {
    rc = raster_batch(m, rreqs);
    // (the rasteriser reads the queries' sensor poses only: readings a caller left to a QueryHook are made now, behind its launches)
    pending_query_hook().run();
    if (rc) {return rc;}
  }

  const double res = m->grid_resolution();
  // Mapper.cpp:577-585
  const double cso = 0.5 * (static_cast<double>(m->side) - 1) * res;
  const double csr = 2 * res;

  auto run = [&](const std::vector<int32_t> & which, double ang_off, double ang_res, bool fine) -> int {
    std::vector<CorrReq> reqs(which.size());
    for (size_t k = 0; k < which.size(); ++k) {
      const int32_t i = which[k];
      CorrReq & q = reqs[k];
      q.slot = i; q.scan = &queries[i];
      if (!fine) {
        std::copy(queries[i].sensor_pose, queries[i].sensor_pose + 3, q.center);
        q.off_x = cso; q.off_y = cso; q.res_x = csr; q.res_y = csr;
      } else {
        std::copy(means + 3 * i, means + 3 * i + 3, q.center);           // Mapper.cpp:625: centre = rMean
        q.off_x = csr * 0.5; q.off_y = csr * 0.5; q.res_x = res; q.res_y = res;   // :622-624
      }
      q.ang_off = ang_off; q.ang_res = ang_res; q.penalize = do_penalize != 0; q.fine = fine;
      std::copy(covs + 9 * i, covs + 9 * i + 9, q.cov);
      q.response = 0; q.status = KH_OK;
    }
    int r = correlate_batch(m, reqs);
    if (r) {return r;}
    for (size_t k = 0; k < which.size(); ++k) {
      const int32_t i = which[k];
      st[i] = reqs[k].status;
      if (reqs[k].status != KH_OK) {continue;}
      std::copy(reqs[k].mean, reqs[k].mean + 3, means + 3 * i);
      std::copy(reqs[k].cov, reqs[k].cov + 9, covs + 9 * i);
      responses[i] = reqs[k].response;
    }
    return KH_OK;
  };
  auto alive = [&](const std::vector<int32_t> & v) {
    std::vector<int32_t> o;
    for (int32_t i : v) {if (st[i] == KH_OK) {o.push_back(i);}}
    return o;
  };

  // coarse search, Mapper.cpp:588-592
  if (!coarse_done) {
    rc = run(active, mp.coarse_search_angle_offset, mp.coarse_angle_resolution, false);
    if (rc) {return rc;}
  }
  // response expansion, Mapper.cpp:594-619
  if (mp.use_response_expansion) {
    std::vector<int32_t> zero;
    for (int32_t i : alive(active)) {if (double_equal(responses[i], 0.0)) {zero.push_back(i);}}
    double newSearchAngleOffset = mp.coarse_search_angle_offset;
    for (uint32_t k = 0; k < 3 && !zero.empty(); ++k) {
      newSearchAngleOffset += 20 * kPi180;
      rc = run(zero, newSearchAngleOffset, mp.coarse_angle_resolution, false);
      if (rc) {return rc;}
      std::vector<int32_t> still;
      for (int32_t i : alive(zero)) {if (double_equal(responses[i], 0.0)) {still.push_back(i);}}
      zero.swap(still);
    }
  }
  // fine search, Mapper.cpp:621-629
  if (do_refine && !fine_done) {
    rc = run(alive(active), 0.5 * mp.coarse_angle_resolution, mp.fine_search_angle_offset, true);
    if (rc) {return rc;}
  }
  int worst = KH_OK;
  for (int32_t i = 0; i < n; ++i) {
    if (status) {status[i] = st[i];}
    if (st[i] != KH_OK) {worst = st[i];}
  }
  return status ? KH_OK : worst;
}

int kh_matcher_match(kh_matcher * m, const kh_scan * query, const kh_scan * base, int32_t n_base,
  int32_t do_penalize, int32_t do_refine, double mean[3], double cov[9], double * response)
{
  if (!m || !query || !mean || !cov || !response || n_base < 0) {return KH_ERR_INVALID_ARG;}
  const int32_t begin[2] = {0, n_base};
  int32_t status = KH_OK;
  int rc = kh_matcher_match_batch(m, 1, query, base, begin, do_penalize, do_refine, mean, cov, response, &status);
  if (rc) {return rc;}
  return status;
}

int kh_matcher_grid_info(kh_matcher * m, int32_t slot, kh_grid_info * out)
{
  if (!m || !out || slot < 0 || slot >= m->max_batch) {return KH_ERR_INVALID_ARG;}
  out->width = m->width; out->height = m->height; out->width_step = m->ws; out->data_size = m->data_size;
  out->roi_x = m->roi_x; out->roi_y = m->roi_y; out->roi_w = m->roi_w; out->roi_h = m->roi_h;
  out->kernel_size = m->kernel_size; out->search_side = m->side;
  out->offset_x = m->slots[slot].off_x; out->offset_y = m->slots[slot].off_y; out->scale = m->scale;
  return KH_OK;
}

int kh_matcher_read_grid(kh_matcher * m, int32_t slot, uint8_t * out)
{
  if (!m || !out || slot < 0 || slot >= m->max_batch) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  KH_HIP(hipStreamSynchronize(m->stream));
  KH_HIP(hipMemcpy(out, m->slots[slot].d_grid, static_cast<size_t>(m->data_size), hipMemcpyDeviceToHost));
  return KH_OK;
}

int kh_matcher_read_kernel(kh_matcher * m, uint8_t * out)
{
  if (!m || !out) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  KH_HIP(hipMemcpy(out, m->d_kernel, m->kernel.size(), hipMemcpyDeviceToHost));
  return KH_OK;
}

int kh_matcher_read_lookup(kh_matcher * m, int32_t slot, int32_t * n_angles, int32_t * n_points, int32_t * out)
{
  if (!m || slot < 0 || slot >= m->max_batch || !n_angles || !n_points) {return KH_ERR_INVALID_ARG;}
  Slot & s = m->slots[slot];
  if (!s.has_last) {return KH_ERR_NOT_FOUND;}
  *n_angles = s.last.na; *n_points = s.last.P;
  if (out) {
    KH_HIP(hipSetDevice(m->device));
    KH_HIP(hipStreamSynchronize(m->stream));
    KH_HIP(hipMemcpy(out, s.d_table, sizeof(int32_t) * s.last.na * s.last.P, hipMemcpyDeviceToHost));
  }
  return KH_OK;
}

int kh_matcher_read_volume(kh_matcher * m, int32_t slot, int32_t * nx, int32_t * ny, int32_t * na,
  int32_t * out_sums, double * out_responses)
{
  if (!m || slot < 0 || slot >= m->max_batch || !nx || !ny || !na) {return KH_ERR_INVALID_ARG;}
  Slot & s = m->slots[slot];
  if (!s.has_last) {return KH_ERR_NOT_FOUND;}
  if (s.volume_stale) {set_error("the last search re-scored an off-lattice best pose: its volume was not kept"); return KH_ERR_NOT_FOUND;}
  const CorrHost & c = s.last;
  *nx = c.nx; *ny = c.ny; *na = c.na;
  if (!out_sums && !out_responses) {return KH_OK;}
  KH_HIP(hipSetDevice(m->device));
  KH_HIP(hipStreamSynchronize(m->stream));
  const size_t plane = static_cast<size_t>(c.nx) * c.ny, vol = plane * c.na;
  if (out_sums) {
    std::vector<int32_t> tmp(vol);
    KH_HIP(hipMemcpy(tmp.data(), s.d_sums, vol * 4, hipMemcpyDeviceToHost));
    for (int32_t a = 0; a < c.na; ++a) {
      for (size_t p = 0; p < plane; ++p) {out_sums[p * c.na + a] = tmp[static_cast<size_t>(a) * plane + p];}
    }
  }
  if (out_responses) {
    if (!m->keep_responses || !s.d_resp) {set_error("response volume not kept: call kh_matcher_set_debug(m, 1) first"); return KH_ERR_NOT_FOUND;}
    std::vector<double> tmp(vol);
    KH_HIP(hipMemcpy(tmp.data(), s.d_resp, vol * 8, hipMemcpyDeviceToHost));
    for (int32_t a = 0; a < c.na; ++a) {
      for (size_t p = 0; p < plane; ++p) {out_responses[p * c.na + a] = tmp[static_cast<size_t>(a) * plane + p];}
    }
  }
  return KH_OK;
}

// ScanMatcher::ComputePositionalCovariance (Mapper.cpp:874-966) on the search-space probabilities the last COARSE
// CorrelateScan of `slot` left behind
int kh_matcher_positional_covariance(kh_matcher * m, int32_t slot, const double best_pose[3], double best_response,
  const double center[3], const double search_offset[2], const double search_resolution[2], double angle_resolution,
  double cov[9])
{
  if (!m || slot < 0 || slot >= m->max_batch || !best_pose || !center || !search_offset || !search_resolution || !cov) {return KH_ERR_INVALID_ARG;}
  if (!(search_resolution[0] > 0.0) || !(search_resolution[1] > 0.0)) {return KH_ERR_INVALID_ARG;}
  Slot & s = m->slots[slot];
  if (!s.has_last_coarse) {set_error("no coarse search has run in this slot"); return KH_ERR_NOT_FOUND;}
  WalkGeometry wg;
  std::copy(center, center + 3, wg.center);
  wg.off_x = search_offset[0]; wg.off_y = search_offset[1]; wg.res_x = search_resolution[0]; wg.res_y = search_resolution[1];
  wg.ang_res = angle_resolution;
  return positional_covariance(m, s.last_coarse, s.last_lattice, wg, best_pose, best_response, cov);
}

// ScanMatcher::ComputeAngularCovariance (Mapper.cpp:977-1025): GetResponse of every search angle at the best pose's cell
// (one 1 x 1 search on the GPU against the grid currently in `slot`), then the reference's accumulation.  Only cov[8]
// (theta-theta) is written.
int kh_matcher_angular_covariance(kh_matcher * m, int32_t slot, const kh_scan * query, const double best_pose[3],
  double best_response, const double center[3], double angle_offset, double angle_resolution, double cov[9])
{
  if (!m || slot < 0 || slot >= m->max_batch || check_scan(query) != KH_OK || query->n == 0 || !best_pose || !center || !cov ||
    angle_resolution == 0.0) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  std::vector<CorrReq> reqs(1);
  CorrReq & q = reqs[0];
  q.slot = slot; q.scan = query;
  q.center[0] = best_pose[0]; q.center[1] = best_pose[1]; q.center[2] = center[2];
  q.off_x = 0.0; q.off_y = 0.0; q.res_x = m->grid_resolution(); q.res_y = m->grid_resolution();
  q.ang_off = angle_offset; q.ang_res = angle_resolution; q.penalize = false; q.fine = true;
  std::fill(q.cov, q.cov + 9, 0.0);
  q.response = 0; q.status = KH_OK;
  int rc = correlate_batch(m, reqs);
  if (rc) {return rc;}
  Slot & s = m->slots[slot];
  const int32_t na = s.last.na;
  std::vector<int32_t> col(static_cast<size_t>(na));
  KH_HIP(hipMemcpy(col.data(), s.d_sums, sizeof(int32_t) * na, hipMemcpyDeviceToHost));
  const double bestAngle = normalize_angle_difference(best_pose[2], center[2]);
  cov[8] = angular_variance(col, s.last.denom, center[2], angle_offset, angle_resolution, bestAngle, best_response);
  return KH_OK;
}

int kh_matcher_seq_stats(kh_matcher * m, int64_t out[8])
{
  if (!m || !out) {return KH_ERR_INVALID_ARG;}
  const int64_t * v = seq_stats(m);
  std::copy(v, v + kSeqStatWords, out);
  return KH_OK;
}

void * kh_matcher_stream(kh_matcher * m) {return m ? reinterpret_cast<void *>(m->stream) : nullptr;}

int kh_matcher_profile(kh_matcher * m, int32_t enable, double * score_ms, int64_t * score_launches,
  double * raster_ms, int64_t * raster_launches)
{
  if (!m) {return KH_ERR_INVALID_ARG;}
  if (score_ms) {*score_ms = m->score_ms;}
  if (score_launches) {*score_launches = m->score_launches;}
  if (raster_ms) {*raster_ms = m->raster_ms;}
  if (raster_launches) {*raster_launches = m->raster_launches;}
  m->profiling = enable != 0;
  m->score_ms = 0; m->raster_ms = 0; m->score_launches = 0; m->raster_launches = 0; m->score_jobs = 0;
  return KH_OK;
}

int kh_matcher_profile_side(kh_matcher * m, double * offsets_ms, double * ties_ms)
{
  if (!m || !offsets_ms || !ties_ms) {return KH_ERR_INVALID_ARG;}
  *offsets_ms = m->offsets_ms; *ties_ms = m->ties_ms;
  m->offsets_ms = 0; m->ties_ms = 0;
  return KH_OK;
}

int kh_matcher_score_loads(kh_matcher * m, int64_t * wave_loads, int32_t reset)
{
  if (!m || !wave_loads) {return KH_ERR_INVALID_ARG;}
  KH_HIP(hipSetDevice(m->device));
  KH_HIP(hipStreamSynchronize(m->stream));
  for (auto & b : m->batch) {KH_HIP(hipStreamSynchronize(b.side));}
  unsigned long long v = 0;
  KH_HIP(hipMemcpy(&v, m->d_load_counter, 8, hipMemcpyDeviceToHost));
  *wave_loads = static_cast<int64_t>(v);
  if (reset) {KH_HIP(hipMemset(m->d_load_counter, 0, 512));}
  return KH_OK;
}

}  // extern "C"
