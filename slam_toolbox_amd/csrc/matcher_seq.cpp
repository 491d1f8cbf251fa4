// Host side of the fused path of ONE MatchScan (ScanMatcher::MatchScan, Mapper.cpp:534-639); the plan is in matcher_seq.hpp, the
// kernels in matcher_seq.hip.  The exact host half (tables with libm, tie averages, covariances) is the general path's own code
// (prepare_job / finalize_job of matcher_host.cpp): what this file changes is WHEN it runs -- the coarse search's tables are made
// while the rasteriser's kernels run, the fine search's for every coarse angle the device may pick -- and how data moves: job
// descriptors as kernel arguments, tables pulled from host-coherent memory by a kernel, results pushed into host-coherent
// memory and announced by a flag.  Whatever the device cannot finish alone (several best poses, response expansion, an
// off-lattice best pose, a degenerate search) is handed back to the general path, pass by pass.
#include "matcher_private.hpp"
#include "matcher_seq.hpp"

#include <immintrin.h>

namespace kh
{

struct SeqState
{
  SeqMid * d_mid = nullptr; int32_t * d_fsum = nullptr;
  RasterJob * d_job = nullptr;
  uint8_t * h_stage = nullptr; uint8_t * d_stage = nullptr; size_t cap_hstage = 0, cap_dstage = 0;
  unsigned long long * h_out = nullptr; unsigned long long * d_out = nullptr; size_t cap_hout = 0, cap_dout = 0;
  SeqFineOut * h_fine = nullptr;
  int32_t * h_flag = nullptr;
  int32_t seq = 0;
  bool ready = false;                     // the one-time blocks below (job, mid, fine sums, host-coherent result and flag) all exist
  bool unavailable = false;               // sticky: this platform hands out no host-coherent memory (kh_matcher_set_debug does not clear it)
  long long * d_dbg = nullptr;            // KH_SEQ_TIMING=1: phase stamps of kseq_bin [0..7], kseq_prep [8..15], the final kernels [16..31]
  double dbg_acc[32] = {0}; long dbg_calls = 0;
  std::vector<uint8_t> fine_scratch;
  int64_t stats[kSeqStatWords] = {0, 0, 0, 0, 0, 0, 0, 0};
};

#define KS_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      set_error(std::string(#call) + ": " + hipGetErrorString(e_));                          \
      return KH_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

void seq_destroy(kh_matcher * m)
{
  SeqState * q = m->seq;
  if (!q) {return;}
  (void)hipFree(q->d_dbg); (void)hipFree(q->d_mid); (void)hipFree(q->d_fsum); (void)hipFree(q->d_job); (void)hipFree(q->d_stage); (void)hipFree(q->d_out);
  if (q->h_stage) {(void)hipHostFree(q->h_stage);}
  if (q->h_out) {(void)hipHostFree(q->h_out);}
  if (q->h_fine) {(void)hipHostFree(q->h_fine);}
  if (q->h_flag) {(void)hipHostFree(q->h_flag);}
  delete q;
  m->seq = nullptr;
}

const int64_t * seq_stats(const kh_matcher * m)
{
  static const int64_t zeros[kSeqStatWords] = {0, 0, 0, 0, 0, 0, 0, 0};
  return m->seq ? m->seq->stats : zeros;
}

template <class T>
static int ensure_coherent(T *& p, size_t & cap, size_t need)
{
  if (need <= cap && p) {return KH_OK;}
  const size_t n = std::max(need, cap + cap / 2);
  if (p) {cap = 0; KS_HIP(hipHostFree(p)); p = nullptr;}
  KS_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), n * sizeof(T), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  cap = n;
  return KH_OK;
}

// extras behind the standard staging layout of the coarse job
struct SeqLayout {size_t xp, yp, heading, fine_cs, fine_ap, fine_dp, total;};
static SeqLayout seq_layout(size_t base, int32_t nx, int32_t ny, int32_t na, int32_t naf)
{
  SeqLayout L;
  size_t o = align_up(base, 16);
  L.xp = o; o = align_up(o + sizeof(double) * nx, 16);
  L.yp = o; o = align_up(o + sizeof(double) * ny, 16);
  L.heading = o; o = align_up(o + sizeof(double) * na, 16);
  L.fine_cs = o; o = align_up(o + sizeof(double) * 2 * na * naf, 16);
  L.fine_ap = o; o = align_up(o + sizeof(double) * na * naf, 16);
  L.fine_dp = o; o = align_up(o + sizeof(double) * 9, 16);
  L.total = align_up(o, 256);
  return L;
}

QueryHook & pending_query_hook()
{
  static thread_local QueryHook hook;
  return hook;
}

void set_pending_query_hook(std::function<void()> fn) {pending_query_hook().fn = std::move(fn);}     // (the mapper: it does not see matcher_private.hpp)
void run_pending_query_hook() {pending_query_hook().run();}

int seq_match(kh_matcher * m, const kh_scan * query, const kh_scan * base, int32_t n_base, bool penalize, bool refine,
  double mean[3], double cov[9], double * response, int * status, bool * coarse_done, bool * fine_done)
{
  *coarse_done = false; *fine_done = false;
  static const bool env_off = std::getenv("KH_SEQ_FUSED") != nullptr && std::atoi(std::getenv("KH_SEQ_FUSED")) == 0;
  if (env_off || m->no_seq) {return KH_OK;}
  if (!m->seq) {m->seq = new SeqState();}
  if (m->seq->unavailable) {return KH_OK;}
  // a call the kernels' fixed-size tables cannot take goes the general way; stats [6] counts them, [7] keeps the last reason
  auto ineligible = [&](int64_t reason) {m->seq->stats[6] += 1; m->seq->stats[7] = reason; return KH_OK;};
  // a platform that does not hand out host-coherent mapped memory keeps the general path for this handle (reason 7)
  // (its own flag: kh_matcher_set_debug rewrites m->no_seq from its bit 7 on every call)
  auto no_coherent_memory = [&]() {
    (void)hipGetLastError();
    m->seq->unavailable = true; (void)hipStreamSynchronize(m->stream); m->slots[0].first_clean = false; return ineligible(7);
  };
  if (m->profiling || m->keep_responses) {return ineligible(1);}
  if (query->n <= 0 || query->n > kSeqMaxReadings) {return ineligible(2);}
  // ---- eligibility: what the kernels' fixed-size tables can take
  int32_t n_scans = 0, max_n = 1;
  int64_t pts = 0;
  bool any_upload = false;
  for (int32_t b = 0; b < n_base; ++b) {
    const kh_scan & sc = base[b];
    if (sc.points_xy == nullptr || sc.n <= 0) {continue;}        // NULL scan: skipped (Mapper.cpp:1039-1041)
    if (sc.n > kSeqMaxReadings) {return ineligible(3);}
    ++n_scans; pts += sc.n; max_n = std::max(max_n, sc.n);
    any_upload = any_upload || sc.device_points_xy == nullptr;
  }
  const int32_t tiles = m->rt_w * m->rt_h, bm_words = m->bm_w * m->bm_h;
  const int32_t n_foot = static_cast<int32_t>(m->footprint100.size()) - 1;
  if (n_scans == 0 || pts <= 0) {return ineligible(4);}
  if (n_scans > kSeqMaxScans || pts > kSeqMaxPoints || tiles > kSeqMaxTiles) {return ineligible(5);}
  size_t bin_lds = seq_bin_lds_bytes(static_cast<int32_t>(pts), n_foot, tiles, bm_words);
  const bool bm_global = bin_lds > 150 * 1024;
  if (bm_global) {bin_lds = seq_bin_lds_bytes(static_cast<int32_t>(pts), n_foot, tiles, 0);}
  if (bin_lds > 150 * 1024) {return ineligible(6);}
  const int32_t np = static_cast<int32_t>(pts);
  SeqState & Q = *m->seq;
  Slot & s = m->slots[0];
  hipStream_t st = m->stream;
  const kh_match_params & mp = m->params;
  int rc = KH_OK;
  // ---- scratch
  const size_t npad = (static_cast<size_t>(np) + 3) & ~static_cast<size_t>(3);
  rc = ensure_device(s.d_ractive, s.cap_ractive, static_cast<size_t>(np), st); if (rc) {return rc;}
  rc = ensure_device(s.d_rlists, s.cap_rlists, npad * 10, st); if (rc) {return rc;}
  if (!Q.ready) {
    // keyed on `ready`, not on the first pointer: a failure half way (the next call would have skipped the block and dereferenced the
    // missing pieces) leaves what it allocated for the retry
    if (!Q.d_job) {KS_HIP(hipMalloc(reinterpret_cast<void **>(&Q.d_job), sizeof(RasterJob)));}
    if (!Q.d_mid) {KS_HIP(hipMalloc(reinterpret_cast<void **>(&Q.d_mid), sizeof(SeqMid)));}
    if (!Q.d_fsum) {KS_HIP(hipMalloc(reinterpret_cast<void **>(&Q.d_fsum), sizeof(int32_t) * kSeqMaxFine));}
    size_t one = Q.h_fine ? 1 : 0;
    rc = ensure_coherent(Q.h_fine, one, 1); if (rc) {return no_coherent_memory();}
    one = Q.h_flag ? 16 : 0;
    rc = ensure_coherent(Q.h_flag, one, 16); if (rc) {return no_coherent_memory();}
    Q.h_flag[0] = 0;
    if (std::getenv("KH_SEQ_TIMING") && !Q.d_dbg) {
      KS_HIP(hipMalloc(reinterpret_cast<void **>(&Q.d_dbg), sizeof(long long) * 32));
      KS_HIP(hipMemset(Q.d_dbg, 0, sizeof(long long) * 32));
    }
    Q.ready = true;
  }
  const bool fused_tiles = m->kernel_size >= 8;
  // scans the caller does not keep on the device: one upload of their points (the matcher's arena, as in the batch path)
  std::vector<const double *> dev_ptr(static_cast<size_t>(n_base), nullptr);
  if (any_upload) {
    size_t arena_points = 0;
    for (int32_t b = 0; b < n_base; ++b) {
      const kh_scan & sc = base[b];
      if (sc.points_xy == nullptr || sc.n <= 0 || sc.device_points_xy) {continue;}
      arena_points += static_cast<size_t>(sc.n);
    }
    rc = ensure_pinned(m->h_arena, m->cap_harena, arena_points * 2, st); if (rc) {return rc;}
    rc = ensure_device(m->d_arena, m->cap_darena, arena_points * 2, st); if (rc) {return rc;}
    KS_HIP(hipStreamSynchronize(st));                 // an earlier call's upload must have left the pinned mirror
    size_t at = 0;
    for (int32_t b = 0; b < n_base; ++b) {
      const kh_scan & sc = base[b];
      if (sc.points_xy == nullptr || sc.n <= 0 || sc.device_points_xy) {continue;}
      std::memcpy(m->h_arena + 2 * at, sc.points_xy, sizeof(double) * 2 * static_cast<size_t>(sc.n));
      dev_ptr[b] = m->d_arena + 2 * at;
      at += static_cast<size_t>(sc.n);
    }
    KS_HIP(hipMemcpyAsync(m->d_arena, m->h_arena, sizeof(double) * 2 * arena_points, hipMemcpyHostToDevice, st));
  }
  // the two searches of the match (Mapper.cpp:577-592, 621-629) and their device scratch -- first: a slot that earns re-pitched
  // copies of its grid gets them here, and the rasteriser's job below must know them
  const double res = m->grid_resolution();
  const double * pose = query->sensor_pose;
  const double cso = 0.5 * (static_cast<double>(m->side) - 1) * res;
  const double csr = 2 * res;
  CorrReq q;
  q.slot = 0; q.scan = query;
  std::copy(pose, pose + 3, q.center);
  q.off_x = cso; q.off_y = cso; q.res_x = csr; q.res_y = csr;
  q.ang_off = mp.coarse_search_angle_offset; q.ang_res = mp.coarse_angle_resolution; q.penalize = penalize; q.fine = false;
  std::copy(cov, cov + 9, q.cov);
  q.response = 0; q.status = KH_OK;
  CorrHost c;
  rc = init_ctx(q, c); if (rc) {return rc;}
  rc = ensure_slot_scratch(m, q, c, false); if (rc) {return rc;}
  CorrReq qf;
  qf.slot = 0; qf.scan = query;
  qf.center[0] = qf.center[1] = qf.center[2] = 0.0;
  qf.off_x = csr * 0.5; qf.off_y = csr * 0.5; qf.res_x = res; qf.res_y = res;
  qf.ang_off = 0.5 * mp.coarse_angle_resolution; qf.ang_res = mp.fine_search_angle_offset; qf.penalize = penalize; qf.fine = true;
  qf.response = 0; qf.status = KH_OK;
  CorrHost cf;
  bool device_fine = refine;
  if (device_fine) {
    if (init_ctx(qf, cf) != KH_OK || cf.nx != 3 || cf.ny != 3 || cf.na * 9 > kSeqMaxFine) {device_fine = false;}
    else {rc = ensure_slot_scratch(m, qf, cf, false); if (rc) {return rc;}}
  }
  // ---- 1. rasteriser: everything it needs travels as kernel arguments
  // MatchScan steps 1-4, Mapper.cpp:543-569
  s.off_x = pose[0] - (0.5 * (m->roi_w - 1) * res);
  s.off_y = pose[1] - (0.5 * (m->roi_h - 1) * res);
  SeqPrepArgs pa;
  fill_raster_job(m, s, pose, np, npad, pa.job);
  rc = ensure_seq_tables(m, s, np, pa.job); if (rc) {return rc;}
  {
    int32_t k = 0, run = 0;
    for (int32_t b = 0; b < n_base; ++b) {
      const kh_scan & sc = base[b];
      if (sc.points_xy == nullptr || sc.n <= 0) {continue;}
      pa.scans[k] = sc.device_points_xy ? sc.device_points_xy : dev_ptr[b];
      pa.prefix[k] = run; run += sc.n; ++k;
    }
    pa.prefix[k] = run;
    for (int32_t r = k; r < kSeqMaxScans; ++r) {pa.scans[r] = nullptr; pa.prefix[r + 1] = run;}
  }
  pa.n_scans = n_scans; pa.max_n = max_n;
  pa.dbg = Q.d_dbg ? Q.d_dbg + 8 : nullptr;
  pa.d_job = Q.d_job; pa.clear_blocks = 128;     // 2048 waves: one tile of the previous match each
  s.first_clean = false;                                  // until the stamping launch has handed the table back (an error in between leaves marks behind)
  launch_seq_prep(pa, st);
  launch_seq_links(Q.d_job, 1, np, st);
  launch_seq_bin(Q.d_job, 1, bin_lds, bm_global ? 1 : 0, Q.d_dbg, st);
  KS_HIP(hipGetLastError());
  // the query's readings, if the caller left them for now (QueryHook): nothing above read them, everything below does
  pending_query_hook().run();

  // ---- 2. the coarse search's host half (tables with libm), while the kernels above run
  const int32_t naf = device_fine ? cf.na : 1;
  const StageLayout L = stage_layout(c.P, c.nx, c.ny, c.na, q.penalize);
  const SeqLayout X = seq_layout(L.total, c.nx, c.ny, c.na, naf);
  const size_t out_words = align_up(kOutHeaderWords + static_cast<size_t>(c.nx) * c.ny, 32);
  rc = ensure_coherent(Q.h_stage, Q.cap_hstage, X.total); if (rc) {return no_coherent_memory();}
  rc = ensure_device(Q.d_stage, Q.cap_dstage, Q.cap_hstage, st); if (rc) {return rc;}
  rc = ensure_coherent(Q.h_out, Q.cap_hout, out_words); if (rc) {return no_coherent_memory();}
  rc = ensure_device(Q.d_out, Q.cap_dout, out_words, st); if (rc) {return rc;}
  JobShape shape;
  prepare_job(m, q, c, L, Q.h_stage, Q.d_stage, Q.d_out, out_words, 1, false, true, shape);
  CorrJob * job = reinterpret_cast<CorrJob *>(Q.h_stage);
  {
    double * xp = reinterpret_cast<double *>(Q.h_stage + X.xp);
    double * yp = reinterpret_cast<double *>(Q.h_stage + X.yp);
    double * heading = reinterpret_cast<double *>(Q.h_stage + X.heading);
    double * fcs = reinterpret_cast<double *>(Q.h_stage + X.fine_cs);
    double * fap = reinterpret_cast<double *>(Q.h_stage + X.fine_ap);
    double * fdp = reinterpret_cast<double *>(Q.h_stage + X.fine_dp);
    std::copy(c.x_poses.begin(), c.x_poses.end(), xp);
    std::copy(c.y_poses.begin(), c.y_poses.end(), yp);
    for (int32_t a = 0; a < c.na; ++a) {
      // the tie average of ONE pose (Mapper.cpp:802-829): atan2 of its heading's sine and cosine
      const double h = normalize_angle(c.angles[a]);
      double sin_h, cos_h;
      ref_sincos(h, &sin_h, &cos_h);
      double thetaX = 0.0, thetaY = 0.0;
      thetaX += cos_h; thetaY += sin_h;
      const int32_t count = 1;
      thetaX /= count; thetaY /= count;
      heading[a] = std::atan2(thetaY, thetaX);
      if (!device_fine) {continue;}
      // the fine search centred there: its angles (Karto.h:6857-6858) and angle penalties (Mapper.cpp:679-682), as prepare_job
      // computes them for a centre heading of heading[a]
      const double centre = heading[a];
      const double startAngle = centre - qf.ang_off;
      for (int32_t k = 0; k < naf; ++k) {
        const double angle = startAngle + static_cast<uint32_t>(k) * qf.ang_res;
        ref_sincos(angle, &fcs[2 * (static_cast<size_t>(a) * naf + k) + 1], &fcs[2 * (static_cast<size_t>(a) * naf + k)]);
        const double squaredAngleDistance = (angle - centre) * (angle - centre);
        double anglePenalty = 1.0 - (kAngleGain * squaredAngleDistance / mp.angle_variance_penalty);
        anglePenalty = anglePenalty > mp.minimum_angle_penalty ? anglePenalty : mp.minimum_angle_penalty;
        fap[static_cast<size_t>(a) * naf + k] = anglePenalty;
      }
    }
    for (int32_t yi = 0; yi < 3; ++yi) {
      for (int32_t xi = 0; xi < 3; ++xi) {
        double distancePenalty = 1.0;
        if (device_fine) {
          const double squaredDistance = cf.x_poses[xi] * cf.x_poses[xi] + cf.y_poses[yi] * cf.y_poses[yi];
          distancePenalty = 1.0 - (kDistanceGain * squaredDistance / mp.distance_variance_penalty);
          distancePenalty = distancePenalty > mp.minimum_distance_penalty ? distancePenalty : mp.minimum_distance_penalty;
        }
        fdp[yi * 3 + xi] = distancePenalty;
      }
    }
  }
  // ---- 3. the stamps (with the tables' way to the device in the same launch), scoring, finalisation on the device
  const size_t plane = static_cast<size_t>(c.nx) * c.ny;
  SeqStageArgs sa;
  sa.h_stage = Q.h_stage; sa.d_stage = Q.d_stage; sa.bytes = X.total;
  sa.sums = s.d_sums; sa.n_sums = plane * c.na; sa.out = Q.d_out; sa.out_words = out_words;
  if (fused_tiles) {
    launch_seq_tile(Q.d_job, 1, m->d_tab, np, tiles, &sa, st);
  } else {
    launch_raster_tiles(Q.d_job, 1, np, tiles, m->d_kernel, m->kernel_size, st);
    launch_seq_stage(Q.d_job, 1, &sa, st);
  }
  KS_HIP(hipGetLastError());
  s.first_clean = true;
  if (s.d_grid2 != nullptr) {launch_repitch(Q.d_job, 1, tiles, st, true);}
  // table + scoring in one launch for every linear lattice (from the grid itself: a slot's copies, if it has any, are not used)
  const bool fused_score = job->linear != 0 && job->lds_path == 0 && (job->sx == 1 || job->sx == 2);
  if (fused_score) {
    launch_seq_score(Q.d_stage, c.na, c.P, c.nx, c.ny, job->sx, pick_ry(c.ny), st);
    Q.stats[kSeqStatFusedScore] += 1;
  } else {
    launch_offsets(Q.d_stage, L.total, 1, c.na, st);
    launch_score(Q.d_stage, L.total, 1, shape.tiles, c.na, shape.sx, shape.ry, st, m->mfma_score);
  }
  launch_seq_cells(Q.d_stage, static_cast<int32_t>(plane), Q.h_out + kOutHeaderWords, st);
  SeqFinalArgs fa;
  std::memset(&fa, 0, sizeof(fa));
  fa.job = Q.d_stage; fa.h_out = Q.h_out; fa.h_fine = Q.h_fine; fa.h_flag = Q.h_flag;
  Q.seq = Q.seq == INT32_MAX ? 1 : Q.seq + 1;
  fa.seq = Q.seq;
  fa.refine = device_fine ? 1 : 0; fa.naf = naf; fa.fine_penalize = penalize ? 1 : 0;
  fa.cx = c.center[0]; fa.cy = c.center[1];
  fa.xp = reinterpret_cast<const double *>(Q.d_stage + X.xp); fa.yp = reinterpret_cast<const double *>(Q.d_stage + X.yp);
  fa.heading = reinterpret_cast<const double *>(Q.d_stage + X.heading);
  fa.fine_cos_sin = reinterpret_cast<const double *>(Q.d_stage + X.fine_cs);
  fa.fine_ang_pen = reinterpret_cast<const double *>(Q.d_stage + X.fine_ap);
  fa.fine_dist_pen = reinterpret_cast<const double *>(Q.d_stage + X.fine_dp);
  for (int32_t k = 0; k < 3; ++k) {fa.fxp[k] = device_fine ? cf.x_poses[k] : 0.0; fa.fyp[k] = device_fine ? cf.y_poses[k] : 0.0;}
  fa.roi_x = m->roi_x; fa.roi_y = m->roi_y;
  fa.fine_table = s.d_table; fa.fine_sums = s.d_sums;
  fa.dbg = Q.d_dbg ? Q.d_dbg + 16 : nullptr;
  fa.mid = Q.d_mid; fa.fsum = Q.d_fsum;
  launch_seq_final(fa, c.P, st);
  KS_HIP(hipGetLastError());
  Q.stats[kSeqStatCalls] += 1;
  // ---- 4. wait for the flag (the kernel's last store, system scope); the stream is asked now and then so that a failed launch
  // cannot hang the caller
  {
    volatile int32_t * flag = Q.h_flag;
    uint64_t spins = 0;
    while (*flag != Q.seq) {
      _mm_pause();
      if ((++spins & 0x3fff) == 0) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) {
          if (*flag == Q.seq) {break;}
          set_error("fused MatchScan: the stream drained without the result flag"); return KH_ERR_HIP;
        }
        if (e != hipErrorNotReady) {set_error(std::string("fused MatchScan: ") + hipGetErrorString(e)); return KH_ERR_HIP;}
      }
    }
  }
  if (Q.d_dbg) {
    // measurement aid: where kseq_bin and kseq_final spend their time (wall_clock64 = 100 MHz), averaged over 64 calls
    long long w[32];
    KS_HIP(hipStreamSynchronize(st));
    KS_HIP(hipMemcpy(w, Q.d_dbg, sizeof(w), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) {Q.dbg_acc[1 + k] += (w[k + 1] - w[k]) * 0.01;}
    Q.dbg_acc[5] += static_cast<double>(w[5]);
    for (int k = 0; k < 3; ++k) {Q.dbg_acc[16 + k] += (w[16 + k + 1] - w[16 + k]) * 0.01;}
    for (int k = 0; k < 6; ++k) {Q.dbg_acc[8 + k] += (w[8 + k + 1] - w[8 + k]) * 0.01;}
    if (++Q.dbg_calls % 64 == 0) {
      const double n = 64.0;
      std::fprintf(stderr, "[kh seq] kseq_prep (first scan) us: load %.1f  next %.1f  reach %.1f  sides+masks %.1f  flags %.1f  cells %.1f;  ", Q.dbg_acc[8] / 64.0,
        Q.dbg_acc[9] / 64.0, Q.dbg_acc[10] / 64.0, Q.dbg_acc[11] / 64.0, Q.dbg_acc[12] / 64.0, Q.dbg_acc[13] / 64.0);
      std::fprintf(stderr, "kseq_bin us: active set %.1f  count %.1f  scan %.1f  fill %.1f  (candidates %.0f);  the end us: kseq_ties %.1f  "
        "kseq_fine (+ boundaries) %.1f  kseq_done %.1f\n", Q.dbg_acc[1] / n, Q.dbg_acc[2] / n, Q.dbg_acc[3] / n, Q.dbg_acc[4] / n, Q.dbg_acc[5] / n,
        Q.dbg_acc[16] / n, Q.dbg_acc[17] / n, Q.dbg_acc[18] / n);
      for (double & v : Q.dbg_acc) {v = 0.0;}
    }
  }
  // ---- 5. finalisation of the coarse pass (tie average, positional covariance: exact host arithmetic)
  ResultView v;
  v.out = Q.h_out; v.small = nullptr; v.device_work = false;
  rc = finalize_job(m, q, c, v);
  if (rc == kNeedGeneric) {Q.stats[kSeqStatCoarseFallback] += 1; return KH_OK;}     // degenerate search: the general path redoes the pass
  if (rc) {return rc;}
  *coarse_done = true;
  *status = q.status;
  if (q.status != KH_OK) {return KH_OK;}
  std::copy(q.mean, q.mean + 3, mean);
  std::copy(q.cov, q.cov + 9, cov);
  *response = q.response;
  if (!refine) {return KH_OK;}
  // response expansion (Mapper.cpp:594-619) re-runs the coarse pass: the general path's business
  if (mp.use_response_expansion && double_equal(*response, 0.0)) {return KH_OK;}
  if (!device_fine || Q.h_fine->valid == 0) {Q.stats[kSeqStatFineFallback] += 1; return KH_OK;}
  // ---- 6. the device's fine pass: accepted iff it searched exactly where the host's coarse result says (bit for bit)
  std::copy(mean, mean + 3, qf.center);
  std::copy(cov, cov + 9, qf.cov);
  rc = init_ctx(qf, cf); if (rc) {return rc;}
  const StageLayout Lf = stage_layout(cf.P, cf.nx, cf.ny, cf.na, qf.penalize);
  if (Q.fine_scratch.size() < Lf.total) {Q.fine_scratch.resize(Lf.total);}
  JobShape fshape;
  prepare_job(m, qf, cf, Lf, Q.fine_scratch.data(), Q.fine_scratch.data(), Q.d_out, out_words, 1, false, true, fshape);
  bool same = std::memcmp(qf.center, Q.h_fine->centre, sizeof(double) * 3) == 0;
  for (int32_t k = 0; k < 3 && same; ++k) {same = cf.bx[k] == Q.h_fine->bx[k] && cf.by[k] == Q.h_fine->by[k];}
  if (!same) {Q.stats[kSeqStatFineMismatch] += 1; return KH_OK;}
  ResultView vf;
  vf.out = Q.h_fine->out; vf.small = Q.h_fine->sums; vf.device_work = false;
  rc = finalize_job(m, qf, cf, vf);
  if (rc == kNeedGeneric) {Q.stats[kSeqStatFineFallback] += 1; return KH_OK;}      // the best fine pose left the lattice
  if (rc) {return rc;}
  *fine_done = true;
  *status = qf.status;
  Q.stats[kSeqStatFineOnDevice] += 1;
  if (qf.status != KH_OK) {return KH_OK;}
  std::copy(qf.mean, qf.mean + 3, mean);
  std::copy(qf.cov, qf.cov + 9, cov);
  *response = qf.response;
  return KH_OK;
}

}  // namespace kh
