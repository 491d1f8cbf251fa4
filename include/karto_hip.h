/*
 * karto_hip.h -- C ABI of libkartohip.so, the MI355X (gfx950) implementation of slam_toolbox's
 * data-parallel hot path:
 *   (A) karto correlative scan matcher   (reference: lib/karto_sdk/src/Mapper.cpp:477-1208,
 *                                          lib/karto_sdk/include/karto_sdk/Mapper.h:1074-1544,
 *                                          Karto.h:6603-6963)
 *   (B) pose-graph SPA solver plugin     (reference: solvers/ceres_solver.cpp, solvers/ceres_utils.h,
 *                                          karto::ScanSolver Mapper.h:954-1066)
 *
 * Everything is POD: plain pointers and sizes, caller-owned buffers, int status returns.  No C++
 * or torch types cross this boundary.  A handle is NOT re-entrant (like karto::ScanMatcher,
 * which keeps per-call state in members, Mapper.cpp:767-772); different handles may be used
 * from different threads.  The thin C++ adaptors that restore the reference's class surface
 * (karto::ScanMatcher / karto::ScanSolver look-alikes) live in include/karto_hip/ and are
 * header-only over this ABI; INTEGRATION.md shows the binding a slam_toolbox maintainer adds.
 *
 * All paths compute on the GPU.  There is no CPU fallback: every entry point that needs the
 * device returns KH_ERR_NO_DEVICE when no gfx950 device / HIP runtime is usable.
 */
#ifndef KARTO_HIP_H_
#define KARTO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KH_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- status codes */
enum {
  KH_OK = 0,
  KH_ERR_INVALID_ARG = 1,   /* ScanMatcher::Create returns NULL (Mapper.cpp:481-493) / smear out of range throws (Mapper.h:1226-1235) */
  KH_ERR_NO_DEVICE = 2,     /* no HIP device: the product has no CPU fallback */
  KH_ERR_HIP = 3,           /* a HIP runtime call failed (kh_last_error() has the text) */
  KH_ERR_SEARCH = 4,        /* the reference throws std::runtime_error (Mapper.cpp:786-796, 828) */
  KH_ERR_NOT_FOUND = 5,     /* unknown node / constraint id (reference logs and returns) */
  KH_ERR_SOLVER = 6,        /* solution not usable (ceres_solver.cpp:249-254): state left unchanged */
  KH_ERR_IO = 7             /* a pose-graph file could not be read / written */
};

KH_API const char * kh_last_error(void);
KH_API int kh_device_count(void);           /* 0 when no GPU is visible */
KH_API const char * kh_version(void);

/* ---------------------------------------------------------------- scans */
/* What the matcher reads from a karto::LocalizedRangeScan (Karto.h:5380-5760):
 *   ranges       GetRangeReadings()                      n doubles (NaN / inf allowed)
 *   points_xy    GetPointReadings(false): the UNFILTERED world points of all n beams as left by
 *                the scan's last Update() (Karto.h:5644-5704), x0,y0,x1,y1,...
 *   sensor_pose  GetSensorPose()                         x, y, heading
 * Buffers are host memory owned by the caller and only read during the call. */
typedef struct kh_scan {
  int32_t n;
  const double * ranges;
  const double * points_xy;
  double sensor_pose[3];
  /* optional (NULL = not resident): the same 2 * n doubles as points_xy in DEVICE memory of the matcher's device
   * (kh_device_malloc / kh_device_upload).  A scan used as a BASE scan (AddScans / MatchScan) is then read where it
   * lies instead of being uploaded with every call -- the scan store of a mapper belongs in HBM (173 MB
   * for 10 k scans); the caller re-uploads when the scan's pose changes (LocalizedRangeScan::Update).  points_xy must
   * still be valid: the query side and the host half read it. */
  const double * device_points_xy;
} kh_scan;

/* helper restating LocalizedRangeScan::Update for callers without karto objects
 * (Karto.h:5644-5704): pt_i = pose + r_i * (cos, sin)(heading + min_angle + i * ang_res) */
KH_API int kh_scan_points(const double * ranges, int32_t n, const double sensor_pose[3],
                          double min_angle, double angular_resolution, double * out_points_xy);

/* ---------------------------------------------------------------- scan matcher (A) */
typedef struct kh_matcher kh_matcher;

/* The eight Mapper parameters ScanMatcher reads through friend access
 * (Mapper.cpp:590-591, 600, 626-627, 674-682), AS STORED by karto::Mapper -- i.e. the two
 * variance penalties are the already-squared values (setters square them, Mapper.cpp:2562-2570). */
typedef struct kh_match_params {
  double coarse_search_angle_offset;   /* m_pCoarseSearchAngleOffset */
  double coarse_angle_resolution;      /* m_pCoarseAngleResolution   */
  double fine_search_angle_offset;     /* m_pFineSearchAngleOffset   */
  int32_t use_response_expansion;      /* m_pUseResponseExpansion    */
  double distance_variance_penalty;    /* m_pDistanceVariancePenalty (squared) */
  double minimum_distance_penalty;     /* m_pMinimumDistancePenalty  */
  double angle_variance_penalty;       /* m_pAngleVariancePenalty (squared) */
  double minimum_angle_penalty;        /* m_pMinimumAnglePenalty     */
} kh_match_params;

/* karto defaults of Mapper::InitializeParameters (Mapper.cpp:2250-2293) */
KH_API void kh_match_params_default(kh_match_params * p);

/* ScanMatcher::Create (Mapper.cpp:477-522).  device = HIP device ordinal; max_batch = number of
 * independent matches one kh_matcher_match_batch call may carry (each owns a correlation grid in
 * HBM: (side + 2*ceil(range/res) + 2*border)^2 bytes). */
KH_API int kh_matcher_create(double search_size, double resolution, double smear_deviation,
                             double range_threshold, int32_t device, int32_t max_batch,
                             kh_matcher ** out);
KH_API void kh_matcher_destroy(kh_matcher * m);
KH_API int kh_matcher_set_params(kh_matcher * m, const kh_match_params * p);

/* ScanMatcher::MatchScan<LocalizedRangeScanVector> (Mapper.cpp:534-639); base scans in container
 * order.  mean = x, y, heading; cov = row-major 3x3; *response in [0, 1]. */
KH_API int kh_matcher_match(kh_matcher * m, const kh_scan * query, const kh_scan * base,
                            int32_t n_base, int32_t do_penalize, int32_t do_refine,
                            double mean[3], double cov[9], double * response);

/* n independent MatchScan calls in one pass (loop-closure candidate batches, BASELINE config 3).
 * base scans of all matches are concatenated; base_begin[i]..base_begin[i+1] delimits match i
 * (n+1 entries).  status[i] is the per-match status. */
KH_API int kh_matcher_match_batch(kh_matcher * m, int32_t n, const kh_scan * queries,
                                  const kh_scan * base, const int32_t * base_begin,
                                  int32_t do_penalize, int32_t do_refine,
                                  double * means /* 3n */, double * covs /* 9n */,
                                  double * responses /* n */, int32_t * status /* n */);

/* ---- the same batch over several devices of ONE process (loop-closure candidate batches shard
 * across the GPUs of a node; replaces nothing in the reference, which walks the candidates one at a time,
 * Mapper.cpp:1500-1561).  One matcher per entry of `devices` (the same device may be listed more than once: independent
 * members sharing a GPU), each driven by its own host thread; candidate i goes to member i % n_members, every member
 * works through its share in chunks of at most max_batch_per_member, and the results come back in candidate order -- what
 * TryCloseLoop's first-acceptance rule needs.  No collective: the matches are independent.
 * base_device_points: NULL, or n_members pointers per base scan, entry [t * n_members + k] = the copy of base[t]'s
 * points_xy in the memory of member k's device (NULL = not resident there: the member uploads the scan itself);
 * kh_scan.device_points_xy is only honoured by a one-member group without such a table. */
typedef struct kh_matcher_group kh_matcher_group;
KH_API int kh_matcher_group_create(double search_size, double resolution, double smear_deviation, double range_threshold,
                                   const int32_t * devices, int32_t n_devices, int32_t max_batch_per_member,
                                   kh_matcher_group ** out);
KH_API void kh_matcher_group_destroy(kh_matcher_group * g);
KH_API int kh_matcher_group_set_params(kh_matcher_group * g, const kh_match_params * p);
KH_API int32_t kh_matcher_group_size(const kh_matcher_group * g);
KH_API kh_matcher * kh_matcher_group_member(kh_matcher_group * g, int32_t index);     /* owned by the group */
KH_API int32_t kh_matcher_group_device(const kh_matcher_group * g, int32_t index);
KH_API int kh_matcher_group_match_batch(kh_matcher_group * g, int32_t n, const kh_scan * queries, const kh_scan * base,
                                        const int32_t * base_begin, const double * const * base_device_points,
                                        int32_t do_penalize, int32_t do_refine, double * means /* 3n */,
                                        double * covs /* 9n */, double * responses /* n */, int32_t * status /* n */);

/* MapperGraph::TryCloseLoop's pair of matches (Mapper.cpp:1515-1549) for n candidate chains at once: the coarse match of
 * queries[i] against base[base_begin[i] .. base_begin[i + 1]) on the loop matcher `coarse` (doPenalize false, doRefineMatch
 * false), the gate (response > minimum_response_coarse, cov(0,0) and cov(1,1) < maximum_variance_coarse: passed[i]), and for
 * the chains that pass the match of the temporary scan at the coarse pose (the same ranges, point readings recomputed with
 * kh_scan_points for min_angle / angular_resolution) on the sequential matcher `fine` (doPenalize false, refined).  The
 * batch is cut into `pieces` (1 = the two batches one after the other, which measures fastest on one GPU: 10.0 ms against
 * 10.4 ms with 4 pieces for 256 chains) and the two matchers work on neighbouring pieces at the same time.  fine_* entries of chains that did not pass are left untouched.  Both matchers need
 * max_batch >= ceil(n / pieces); their base scans' device_points_xy, if any, must live on BOTH matchers' device. */
KH_API int kh_loop_closure_batch(kh_matcher * coarse, kh_matcher * fine, int32_t n, const kh_scan * queries, const kh_scan * base,
                                 const int32_t * base_begin, double min_angle, double angular_resolution,
                                 double minimum_response_coarse, double maximum_variance_coarse, int32_t pieces,
                                 double * coarse_means, double * coarse_covs, double * coarse_responses, int32_t * passed,
                                 double * fine_means, double * fine_covs, double * fine_responses);

/* MatchScan steps 1-4 + AddScans only (Mapper.cpp:543-574): centre the grid of batch slot
 * `slot` on the query's sensor pose and rasterise the base scans into it. */
KH_API int kh_matcher_add_scans(kh_matcher * m, int32_t slot, const kh_scan * query,
                                const kh_scan * base, int32_t n_base);

/* ScanMatcher::CorrelateScan (Mapper.cpp:712-862) against the grid currently rasterised in
 * `slot`.  cov is in/out: the fine pass (doing_fine_match) only rewrites cov[8]. */
KH_API int kh_matcher_correlate(kh_matcher * m, int32_t slot, const kh_scan * query,
                                const double center[3], const double search_offset[2],
                                const double search_resolution[2], double angle_offset,
                                double angle_resolution, int32_t do_penalize,
                                int32_t doing_fine_match, double mean[3], double cov[9],
                                double * response);

/* The same CorrelateScan on n slots at once (one launch per kernel for the whole batch);
 * arrays are per-slot (slot i uses queries[i], centers[3i..], ...). */
KH_API int kh_matcher_correlate_batch(kh_matcher * m, int32_t n, const kh_scan * queries,
                                      const double * centers, const double search_offset[2],
                                      const double search_resolution[2], double angle_offset,
                                      double angle_resolution, int32_t do_penalize,
                                      int32_t doing_fine_match, double * means, double * covs,
                                      double * responses, int32_t * status);

/* ScanMatcher::ComputePositionalCovariance (Mapper.cpp:874-966): walks the search-space probabilities the last COARSE
 * CorrelateScan of `slot` left behind (the reference keeps them in m_pSearchSpaceProbs) with the caller's geometry. */
KH_API int kh_matcher_positional_covariance(kh_matcher * m, int32_t slot, const double best_pose[3], double best_response,
                                            const double center[3], const double search_offset[2],
                                            const double search_resolution[2], double angle_resolution, double cov[9]);
/* ScanMatcher::ComputeAngularCovariance (Mapper.cpp:977-1025) for `query` against the grid in `slot`: only cov[8] is
 * written.  (The reference reads the lookup table of its last CorrelateScan; here the scan is named.) */
KH_API int kh_matcher_angular_covariance(kh_matcher * m, int32_t slot, const kh_scan * query, const double best_pose[3],
                                         double best_response, const double center[3], double angle_offset,
                                         double angle_resolution, double cov[9]);

/* ---- introspection used by the parity tests and the bench (not needed by the adaptor) ---- */
typedef struct kh_grid_info {
  int32_t width, height, width_step, data_size;      /* Grid<kt_int8u> incl. border (Karto.h:4636-4664) */
  int32_t roi_x, roi_y, roi_w, roi_h;                /* CorrelationGrid ROI (Mapper.h:1204) */
  int32_t kernel_size;                               /* Mapper.h:1240 */
  int32_t search_side;                               /* side of the search-space-probs grid (Mapper.cpp:498) */
  double offset_x, offset_y, scale;                  /* CoordinateConverter of `slot` */
} kh_grid_info;
KH_API int kh_matcher_grid_info(kh_matcher * m, int32_t slot, kh_grid_info * out);
KH_API int kh_matcher_read_grid(kh_matcher * m, int32_t slot, uint8_t * out /* data_size */);
KH_API int kh_matcher_read_kernel(kh_matcher * m, uint8_t * out /* kernel_size^2 */);
/* last lookup table of `slot`: n_angles x n_points int32 (Karto.h:6797-6894); pass out=NULL to query sizes */
KH_API int kh_matcher_read_lookup(kh_matcher * m, int32_t slot, int32_t * n_angles,
                                  int32_t * n_points, int32_t * out);
/* last response volume of `slot` in the reference's order ((y*nX + x)*nAngles + a), Mapper.cpp:688:
 * raw integer sums (GetResponse numerator, Mapper.cpp:1200) and the penalised responses */
KH_API int kh_matcher_read_volume(kh_matcher * m, int32_t slot, int32_t * nx, int32_t * ny,
                                  int32_t * na, int32_t * out_sums, double * out_responses);
/* bit 0: keep the penalised response volume of every CorrelateScan on the device so that
 * kh_matcher_read_volume can return it (parity tests); bit 1: score EVERY search the LDS-staged kernels can take through
 * them (by default only the large ones: windows of at most 61 bytes x 64 rows with >= 1e8 lookups per search, where they
 * were measured faster); bit 6: none (the windowed kernel scores everything); bit 2: dense scoring --
 * do not leave out the beams whose whole search window lies in grid blocks no scan point was stamped into
 * (they add 0 to every pose, so the results are identical either way; for measurements); bit 3: send every batch
 * of >= 128 searches through the chunked pipeline, which otherwise only large searches take (tests); bit 4: score from the
 * grid itself instead of its re-pitched copies (same results; for measurements); bit 5: the windowed kernel takes the byte
 * sums of one-cell searches on the matrix cores (v_mfma_i32_16x16x32_i8) instead of the vector ALU (same results, same speed
 * within 5 %: DESIGN.md section 4); bit 7: kh_matcher_match takes the general (batch) path instead of the fused path of one
 * MatchScan (same results; the parity tests compare the two).  Results are identical under every combination. */
KH_API int kh_matcher_set_debug(kh_matcher * m, int32_t flags);
/* Counters of the fused path ONE MatchScan takes (kh_matcher_match, Mapper.cpp:534-639; csrc/matcher_seq.cpp) since the handle
 * was made: [0] calls that took it, [1] fine passes finished on the device (the coarse pass had exactly one best pose),
 * [2] fine passes handed to the general path (several best poses, response expansion, an off-lattice best pose),
 * [3] fine passes of the device rejected by the host's check of their centre (must stay 0), [4] coarse passes redone by the
 * general path (degenerate searches: more ties than the result block holds), [5] coarse passes scored by the fused
 * table + scoring kernel (linear lattices), [6] calls that went the general way because the fused kernels' fixed-size tables
 * cannot take them, [7] the last such call's reason (1 profiling / kept volume, 2 query beams, 3 readings per base scan,
 * 4 no base readings, 5 scans / points / tiles, 6 LDS, 7 no host-coherent memory: the handle keeps the general path). */
KH_API int kh_matcher_seq_stats(kh_matcher * m, int64_t out[8]);
/* The handle's main HIP stream (hipStream_t as void*): every kernel of a call that is not a chunked batch is launched on it, so
 * the caller can bracket launches with HIP events there; chunked batches (>= 128 large searches) run their chunks on two
 * side streams of the handle that are joined to this stream before the call returns. */
KH_API void * kh_matcher_stream(kh_matcher * m);
/* accumulated GPU time of the scoring kernel (K3) since the last reset, measured with HIP events
 * on the handle's stream when profiling is enabled: total ms and launch count */
KH_API int kh_matcher_profile(kh_matcher * m, int32_t enable, double * score_ms, int64_t * score_launches,
                              double * raster_ms, int64_t * raster_launches);

/* GPU time of the kernels either side of the scoring kernel in the same launches (HIP events on their stream while profiling is
 * enabled): the table / list kernel (K2, K2') and the tie kernel (K4), total ms since the last call (which this one resets).
 * Call it BEFORE kh_matcher_profile(m, 0, ...) reads and resets the launch count. */
KH_API int kh_matcher_profile_side(kh_matcher * m, double * offsets_ms, double * ties_ms);

/* wave-level dword-load instructions (256 B each: 16 lanes x 4 B across, 4 grid rows down) the scoring kernel issued
 * for the searches run while profiling was enabled, tallied on the device by K2 from the beam lists it hands to K3
 * (slow-path beams, which need the per-pose range check, are not included).  This is the L1 (TCP) side of the
 * roofline: bytes = 256 * wave_loads.  reset != 0 zeroes the tally after reading it. */
KH_API int kh_matcher_score_loads(kh_matcher * m, int64_t * wave_loads, int32_t reset);

/* ---------------------------------------------------------------- SPA solver (B) */
typedef struct kh_spa kh_spa;

/* options hard-wired by CeresSolver::Configure (ceres_solver.cpp:157-186) + Ceres defaults */
typedef struct kh_spa_options {
  int32_t max_num_iterations;          /* Ceres default 50 */
  double function_tolerance;           /* 1e-3 */
  double gradient_tolerance;           /* 1e-6 */
  double parameter_tolerance;          /* 1e-3 */
  double min_relative_decrease;        /* 1e-3 */
  double initial_trust_region_radius;  /* 1e4 */
  double max_trust_region_radius;      /* 1e8 */
  double min_trust_region_radius;      /* 1e-16 */
  double min_lm_diagonal;              /* 1e-6 */
  double max_lm_diagonal;              /* 1e32 */
  int32_t max_num_consecutive_invalid_steps;   /* 3 */
  int32_t use_nonmonotonic_steps;              /* true */
  int32_t max_consecutive_nonmonotonic_steps;  /* 3 */
  int32_t jacobi_scaling;                      /* true */
  /* `ceres_loss_function` (ceres_solver.cpp:60-94): KH_LOSS_NONE (squared loss, the default), KH_LOSS_HUBER =
   * ceres::HuberLoss(loss_scale), KH_LOSS_CAUCHY = ceres::CauchyLoss(loss_scale); the reference hard-wires
   * the scale 0.7 for both */
  int32_t loss_function;
  double loss_scale;                           /* 0.7 */
} kh_spa_options;
enum { KH_LOSS_NONE = 0, KH_LOSS_HUBER = 1, KH_LOSS_CAUCHY = 2 };
KH_API void kh_spa_options_default(kh_spa_options * o);

typedef struct kh_spa_summary {
  int32_t iterations;           /* LM iterations executed (successful + unsuccessful) */
  int32_t successful_steps;
  int32_t termination;          /* 0 convergence, 1 no convergence (max iters), 2 failure */
  int32_t usable;               /* Summary::IsSolutionUsable() */
  double initial_cost, final_cost;
  double linearize_ms, solve_ms, total_ms;   /* GPU/host wall split of Compute() */
  int64_t nnz_factor;           /* scalar non-zeros of the Cholesky factor */
  /* measurement (HIP events on the solver's stream, summed over the LM iterations of this Compute()) */
  int64_t factor_flops;         /* floating-point operations of ONE numeric factorisation: sum over the fronts of
                                   sum_{j < ns} (m - j)^2 (partial dense Cholesky of ns pivots of an m x m front) */
  int32_t factorizations;       /* numeric factorisations executed (= LM iterations) */
  int32_t levels;               /* elimination-tree levels = dependent launches per factorisation */
  double factor_gpu_ms;         /* assemble + factor + forward sweeps */
  double backward_gpu_ms;       /* backward sweeps + step evaluation */
  double linearize_gpu_ms;      /* edge linearisation + gathers of H and g (every evaluation point) */
  double symbolic_ms;           /* host: pattern + ordering + symbolic factorisation + uploads (0 when the topology was cached) */
  double worst_linear_residual; /* KH_SPA_CHECK=1 only (else 0): max over the iterations of |(Hs + D/radius) step + gs| / |gs|,
                                   evaluated from the block-sparse matrix, independent of the factorisation */
  int32_t analysis;             /* symbolic analysis of this Compute(): 0 none (topology unchanged), 1 full nested dissection,
                                   2 incremental (supernodes of the last dissection reused, new nodes as leading leaves) */
  int32_t analysis_pad;
} kh_spa_summary;

KH_API int kh_spa_create(int32_t device, kh_spa ** out);
/* Test / measurement switches, 0 = none.  Bit 0: every LM iteration also evaluates the residual of its linear solve from
 * the block-sparse matrix (kh_spa_summary.worst_linear_residual).  Bit 1: HIP events around the phases of every iteration
 * (kh_spa_summary.factor_gpu_ms, backward_gpu_ms, linearize_gpu_ms; 0 without it -- each event costs the stream 5-6 us).  Bits 4-7: numeric factorisation kernels -- 0 default
 * (3), 3 level pipeline potrf/trsm/syrk, 2 panel-pair kernels, 1 their first form; all give the same factor.  How a front's
 * update matrix reaches its parent in the level pipeline: added into the parent front by the kernel that computes it
 * (default); bit 8: every front reads its children's update matrices in place; bit 9: an extend-add launch per level sums
 * them in (rounds 3-5). */
KH_API int kh_spa_set_debug(kh_spa * s, int32_t flags);
/* Multi-GPU (one process per GPU, every rank holds the same graph): rank r linearises the edge block
 * [E*r/world, E*(r+1)/world) into PARTIAL normal equations, and `allreduce` -- supplied by the host
 * framework, e.g. RCCL through torch.distributed -- must sum `count` doubles at `device_buf` (H followed
 * by g, one contiguous buffer) in place across the ranks, ordered after the work already queued on
 * `hip_stream` (a hipStream_t) and complete, as far as that stream is concerned, when it returns 0.  The
 * factorisation and the LM control stay replicated.  world = 1 (default) disables sharding. */
typedef int (*kh_allreduce_fn)(void * user, double * device_buf, int64_t count, void * hip_stream);
KH_API int kh_spa_set_sharding(kh_spa * s, int32_t rank, int32_t world, kh_allreduce_fn allreduce, void * user);
/* The same sharding with the collective INSIDE the library: ncclAllReduce(sum, f64) of RCCL on `comm` (see kh_comm_*
 * below), rank and world taken from the communicator, which must live on the solver's device and outlive the solver's
 * use of it.  comm = NULL returns to the unsharded solver.  With a one-rank communicator the (identity) all-reduce is
 * still issued, so a single-GPU box exercises the whole path. */
typedef struct kh_comm kh_comm;
KH_API int kh_spa_set_comm(kh_spa * s, kh_comm * comm);
KH_API void kh_spa_destroy(kh_spa * s);
KH_API int kh_spa_set_options(kh_spa * s, const kh_spa_options * o);
KH_API int kh_spa_reset(kh_spa * s);                                   /* ScanSolver::Reset  (ceres_solver.cpp:279-314) */
KH_API int kh_spa_clear(kh_spa * s);                                   /* ScanSolver::Clear  (ceres_solver.cpp:272-276) */
KH_API int kh_spa_add_node(kh_spa * s, int32_t id, const double pose[3]);   /* AddNode (ceres_solver.cpp:317-336) */
/* AddConstraint (ceres_solver.cpp:339-392): z = LinkInfo::GetPoseDifference, cov = LinkInfo::GetCovariance
 * (row-major 3x3); the inverse (Karto.h:2533-2577), symmetrisation and upper Cholesky happen inside */
KH_API int kh_spa_add_constraint(kh_spa * s, int32_t id_a, int32_t id_b, const double z[3],
                                 const double cov[9]);
KH_API int kh_spa_remove_node(kh_spa * s, int32_t id);                 /* ceres_solver.cpp:395-427 */
KH_API int kh_spa_remove_constraint(kh_spa * s, int32_t id_a, int32_t id_b);  /* :430-448 */
KH_API int kh_spa_modify_node(kh_spa * s, int32_t id, const double pose[3]);  /* :451-461 (adds old yaw) */
KH_API int kh_spa_get_node(kh_spa * s, int32_t id, double pose[3]);    /* getGraph()/GetNodeOrientation */
KH_API int32_t kh_spa_num_nodes(kh_spa * s);
KH_API int32_t kh_spa_num_constraints(kh_spa * s);
KH_API int kh_spa_compute(kh_spa * s, kh_spa_summary * summary);       /* Compute (ceres_solver.cpp:214-269) */
/* Trace of the last Compute(), one row of 8 doubles per trust-region iteration -- what ceres::IterationSummary holds for it, so
 * that a Ceres run of the same problem (oracle/ceres_driver.cpp, where Ceres is installed) can be laid beside it line by line:
 * [0] iteration (1-based), [1] cost of the iterate the step starts from, [2] cost of the candidate, [3] model cost change,
 * [4] trust-region radius the step was computed with, [5] radius after the iteration's update, [6] step norm (scaled space),
 * [7] verdict: 1 accepted, 0 rejected, -1 invalid step, 2 / 3 terminated on parameter / function tolerance.
 * Writes min(capacity, rows) rows, *n_rows = rows available. */
KH_API int kh_spa_iteration_log(kh_spa * s, int32_t capacity, double * rows, int32_t * n_rows);
/* GetCorrections (ceres_solver.cpp:272): pass ids=NULL to query the count */
KH_API int kh_spa_get_corrections(kh_spa * s, int32_t * n, int32_t * ids, double * poses /* 3n */);
/* ---- pose-graph files.  The reference persists a Boost binary archive of the whole
 * Mapper (Mapper.cpp:2635-2651, serialization.hpp:38-82; not readable without Boost) and rebuilds the solver from it
 * with Reset / AddNode* / AddConstraint* (slam_toolbox_common.cpp:959-1016).  Here the solver's own state is the file:
 *   text    g2o SE2 records   VERTEX_SE2 id x y theta            (AddNode order; the first one is the gauge)
 *                             FIX id                             (optional; must name the first vertex)
 *                             EDGE_SE2 a b dx dy dtheta  i00 i01 i02 i11 i12 i22   (upper triangle of the information)
 *           '#' comments and blank lines are skipped; numbers are written with 17 significant digits, so a
 *           save / load round trip is bit-exact;
 *   binary  "KHPG\1\0\0\0", int64 n, int64 m, int32 id[n], f64 pose[3n], int32 a[m], int32 b[m], f64 z[3m],
 *           f64 info[6m], little endian.
 * kh_spa_load parses and validates the whole file first (KH_ERR_IO / KH_ERR_INVALID_ARG leave the current graph in
 * place), then Reset + AddNode + AddConstraint in file order; the format is detected from the first 8 bytes. */
enum { KH_GRAPH_TEXT = 0, KH_GRAPH_BINARY = 1 };
KH_API int kh_spa_save(kh_spa * s, const char * path, int32_t format);
KH_API int kh_spa_load(kh_spa * s, const char * path);
/* AddConstraint for callers that hold the information matrix (upper triangle 00 01 02 11 12 22, what EDGE_SE2
 * carries) instead of LinkInfo's covariance: skips the inverse of ceres_solver.cpp:364-375, same llt() after it */
KH_API int kh_spa_add_constraint_information(kh_spa * s, int32_t id_a, int32_t id_b, const double z[3],
                                             const double info_upper[6]);
/* enumeration in insertion order (what the files hold); KH_ERR_NOT_FOUND past the end */
KH_API int kh_spa_get_node_at(kh_spa * s, int32_t index, int32_t * id, double pose[3]);
/* all nodes at once, insertion order: ids[kh_spa_num_nodes], poses[3 * kh_spa_num_nodes] (either may be NULL) */
KH_API int kh_spa_get_nodes(kh_spa * s, int32_t * ids, double * poses);
KH_API int kh_spa_get_constraint(kh_spa * s, int32_t index, int32_t * id_a, int32_t * id_b, double z[3],
                                 double info_upper[6]);
/* LinkInfo::Update (Mapper.h:174-188) for callers without karto objects */
KH_API int kh_link_info(const double pose1[3], const double pose2[3], const double cov[9],
                        double pose_difference[3], double cov_out[9]);

/* ---------------------------------------------------------------- multi-GPU communicator (RCCL over xGMI) */
/* One process per GPU.  Rank 0 makes an id with kh_comm_unique_id and hands the 128 bytes to the other ranks by whatever
 * the launcher offers (MPI, a torch.distributed store, a file); every rank then calls kh_comm_create with its device,
 * its rank and the world size (collective: returns when all ranks have joined, like ncclCommInitRank).  librccl is
 * bound at run time, so single-GPU users never load it; KH_ERR_NO_DEVICE when it cannot be found. */
#define KH_COMM_ID_BYTES 128
KH_API int kh_comm_unique_id(uint8_t id[KH_COMM_ID_BYTES]);
KH_API int kh_comm_create(int32_t device, int32_t rank, int32_t world, const uint8_t id[KH_COMM_ID_BYTES], kh_comm ** out);
KH_API void kh_comm_destroy(kh_comm * c);
KH_API int32_t kh_comm_rank(const kh_comm * c);
KH_API int32_t kh_comm_world(const kh_comm * c);
KH_API int32_t kh_comm_device(const kh_comm * c);     /* the device the communicator was created on; -1 for NULL */
/* in-place sum of `count` doubles at device_buf across the ranks, enqueued on hip_stream (a hipStream_t) */
KH_API int kh_comm_allreduce_sum_f64(kh_comm * c, double * device_buf, int64_t count, void * hip_stream);
/* every rank contributes count_per_rank doubles; device_recv (world * count_per_rank) holds them in rank order.  What the
 * sharded candidate matcher uses to collect the 13 result doubles of every pair */
KH_API int kh_comm_allgather_f64(kh_comm * c, const double * device_send, double * device_recv, int64_t count_per_rank,
                                 void * hip_stream);

/* device buffers for callers without a HIP binding of their own (the two collectives take device pointers); download
 * waits for all work queued on the device first */
KH_API int kh_device_malloc(int32_t device, int64_t bytes, void ** out);
KH_API void kh_device_free(void * p);
KH_API int kh_device_upload(void * device_dst, const void * host_src, int64_t bytes);
/* the upload queued on a HIP stream (hipStream_t as void *, e.g. kh_matcher_stream): ordered in front of whatever is launched there next */
KH_API int kh_device_upload_on(void * device_dst, const void * host_src, int64_t bytes, void * hip_stream);
KH_API int kh_device_download(void * host_dst, const void * device_src, int64_t bytes);
/* self-test of the once-per-device bookkeeping behind the kernels' dynamic-LDS attribute (csrc/lds_attr.hpp), run on made-up
 * device ids; 0 = every check holds.  Needs no device: the CPU test suite calls it. */
KH_API int kh_selftest_lds_attr(void);

/* ---------------------------------------------------------------- loop-candidate enumeration (next row f-1) */
/* GPU-resident copy of what karto::MapperGraph's candidate search reads: the reference position
 * GetReferencePose(use_scan_barycenter) of every scan of a sensor in scan-list order (NULL scans left out:
 * the reference skips them, Mapper.cpp:1980-1982) and the adjacency of the pose graph in CSR form with the
 * neighbours in Vertex::GetAdjacentVertices order (Mapper.h:338-361).  Indices are positions in that list. */
typedef struct kh_graph kh_graph;
KH_API int kh_graph_create(int32_t device, kh_graph ** out);
KH_API void kh_graph_destroy(kh_graph * g);
KH_API int kh_graph_set(kh_graph * g, int32_t n_scans, const double * ref_xy /* 2n */,
                        const int32_t * adj_ptr /* n+1 */, const int32_t * adj_idx);
KH_API int kh_graph_set_positions(kh_graph * g, int32_t n_scans, const double * ref_xy);   /* after CorrectPoses */
/* For every query scan: all the chains successive MapperGraph::FindPossibleLoopClosure calls return
 * (Mapper.cpp:1960-2010, enumerated like TryCloseLoop does, Mapper.cpp:1500-1560), FindNearLinkedScans
 * (Mapper.cpp:1795-1806) included, for the CURRENT graph state (speculative batch).
 * Chains are runs of consecutive scans: chains[2k], chains[2k+1] = first, last index; query i owns
 * chains chain_begin[i] .. chain_begin[i+1]-1 (n_queries+1 entries).  *n_chains is the total even when it
 * exceeds cap_chains (then only the first cap_chains are written). */
KH_API int kh_graph_find_loop_candidates(kh_graph * g, int32_t n_queries, const int32_t * query_scans,
                                         double loop_search_maximum_distance, int32_t loop_match_minimum_chain_size,
                                         int32_t * chain_begin, int32_t * chains, int32_t cap_chains,
                                         int32_t * n_chains);
/* The same with TryCloseLoop's resume index: query i enumerates as successive FindPossibleLoopClosure calls starting at
 * rStartNum = start_scans[i] do (Mapper.cpp:1963, 1976: the chain under construction is empty at the resume point, so a
 * run of candidate scans that straddles it counts from there).  What the speculative batch re-issues after a closure
 * moved the poses.  start_scans = NULL: all zero. */
KH_API int kh_graph_find_loop_candidates_from(kh_graph * g, int32_t n_queries, const int32_t * query_scans,
                                              const int32_t * start_scans, double loop_search_maximum_distance,
                                              int32_t loop_match_minimum_chain_size, int32_t * chain_begin, int32_t * chains,
                                              int32_t cap_chains, int32_t * n_chains);
KH_API double kh_graph_last_kernel_ms(kh_graph * g);
/* After scans were removed (lifelong mode) the reference's candidate walks stop at the SIZE of its scan map, which has
 * fallen behind the largest scan id (Mapper.cpp:1974-1976, 1751-1756): only the first n_visit scans of the list are
 * visited as chain members (all of them still count for the breadth-first "linked" test).  kh_graph_set resets it to n. */
KH_API int kh_graph_set_scan_limit(kh_graph * g, int32_t n_visit);
/* Incremental edits of the store (what MapperGraph::AddVertex / AddEdge and a scan's SetSensorPose do to the reference's
 * graph): a mapper that appends a scan and links it a few times per processed scan does not rebuild the store with
 * kh_graph_set every time.  Positions are list positions as in kh_graph_set; an edge goes to the END of both
 * adjacency lists (Vertex::GetAdjacentVertices order).  Removing a scan renumbers the list: rebuild with kh_graph_set. */
KH_API int kh_graph_append_scan(kh_graph * g, const double ref_xy[2]);
KH_API int kh_graph_add_edge(kh_graph * g, int32_t scan_a, int32_t scan_b);
KH_API int kh_graph_set_position(kh_graph * g, int32_t scan, const double ref_xy[2]);
/* MapperGraph::FindNearLinkedVertices (Mapper.cpp:1808-1819): the vertices a breadth-first traversal from the scan
 * reaches through vertices within max_distance of it, in visiting order (the scan itself first).  *n_found is the total. */
KH_API int kh_graph_find_near_linked(kh_graph * g, int32_t query_scan, double max_distance, int32_t * scans, int32_t cap,
                                     int32_t * n_found);
/* The rest of the row -- neighbourhood-sized, exact host arithmetic, no kernel:
 * MapperGraph::FindNearChains (Mapper.cpp:1683-1793) for one scan of the current graph: the maximal runs of
 * consecutive scans within link_scan_maximum_distance of it that hold a near linked scan (FindNearLinkedScans,
 * Mapper.cpp:1795-1806), in the order the breadth-first traversal meets them, without the run that contains the scan
 * itself.  chains[2k], chains[2k+1] = first, last index; *n_chains is the total even beyond cap_chains. */
KH_API int kh_graph_find_near_chains(kh_graph * g, int32_t query_scan, double link_scan_maximum_distance,
                                     int32_t * chains, int32_t cap_chains, int32_t * n_chains);
/* MapperGraph::GetClosestScanToPose (Mapper.cpp:1563-1582): first scan of the list with the smallest squared
 * distance of its reference position to pose_xy; -1 for an empty list */
KH_API int kh_graph_closest_scan_to_pose(kh_graph * g, const int32_t * scans, int32_t n, const double pose_xy[2],
                                         int32_t * closest);
/* MapperGraph::ComputeWeightedMean (Mapper.cpp:1914-1958): inverse-covariance weighted mean of n poses (means 3n,
 * covariances 9n row-major), heading = atan2 of the mean sine and cosine */
KH_API int kh_weighted_mean(int32_t n, const double * means, const double * covariances, double mean[3]);

/* ---------------------------------------------------------------- occupancy grid (next row f-2) */
/* karto::OccupancyGrid::CreateFromScans (Karto.h:5947-5962, 6118-6274).  Scans are handed over like to the
 * matcher (ranges + UNFILTERED point readings + sensor pose, GetPointReadings(false) at Karto.h:6157). */
typedef struct kh_occupancy kh_occupancy;
/* OccupancyGrid::ComputeDimensions (Karto.h:6086-6112) from the scans' bounding boxes (Karto.h:5694-5700) */
KH_API int kh_occupancy_compute_dimensions(int32_t n_scans, const kh_scan * scans, double min_range,
                                           double range_threshold, double resolution, int32_t * width,
                                           int32_t * height, double offset[2]);
KH_API int kh_occupancy_create(int32_t width, int32_t height, double offset_x, double offset_y, double resolution,
                               int32_t device, kh_occupancy ** out);
KH_API void kh_occupancy_destroy(kh_occupancy * g);
KH_API int kh_occupancy_clear(kh_occupancy * g);
/* AddScan for every scan (Karto.h:6148-6189): pass / hit counters only */
KH_API int kh_occupancy_add_scans(kh_occupancy * g, int32_t n_scans, const kh_scan * scans, double range_threshold,
                                  double min_range, double max_range);
/* Update (Karto.h:6257-6274); karto defaults: min_pass_through 2, occupancy_threshold 0.1 (Karto.h:5920-5921) */
KH_API int kh_occupancy_update(kh_occupancy * g, uint32_t min_pass_through, double occupancy_threshold);
/* cells: width_step * height bytes (0 unknown, 100 occupied, 255 free, Karto.h:4379-4381); pass / hits: the
 * counter grids (same layout, uint32); any pointer may be NULL */
KH_API int kh_occupancy_read(kh_occupancy * g, uint8_t * cells, uint32_t * pass, uint32_t * hits);
KH_API int kh_occupancy_info(kh_occupancy * g, int32_t * width, int32_t * height, int32_t * width_step,
                             double * trace_ms, int64_t * beams_traced);

/* ---------------------------------------------------------------- lifelong node-decay scoring (next row f-4) */
/* LifelongSlamToolbox::computeScores (src/experimental/slam_toolbox_lifelong.cpp:295-329) with the metrics
 * of :373-478 and the objective of :199-250.  A kh_scan_box is what the scoring reads from a scan / vertex:
 * GetBarycenterPose(), GetBoundingBox().GetSize(), the FILTERED point readings GetPointReadings(true), the
 * unique id, the vertex's edge count and its current score. */
typedef struct kh_scan_box {
  double barycenter[2];
  double bbox_size[2];            /* width, height */
  int32_t unique_id;
  int32_t n_edges;                /* Vertex::GetEdges().size() */
  double score;                   /* Vertex::GetScore() */
  int32_t n_points;
  const double * points_xy;       /* filtered point readings, host memory */
} kh_scan_box;
typedef struct kh_decay_params {
  double iou_thresh;              /* lifelong_minimum_score          0.10 */
  double iou_match;               /* lifelong_iou_match              0.85 */
  double removal_score;           /* lifelong_node_removal_score     0.10 (used by the caller, :166) */
  double overlap_scale;           /* lifelong_overlap_score_scale    0.5  */
  double constraint_scale;        /* lifelong_constraint_multiplier  0.05 */
  double nearby_penalty;          /* lifelong_nearby_penalty         0.001 */
  double candidates_scale;        /* lifelong_candidates_scale       0.03 (computed but unused upstream, :231-240) */
  int32_t scan_buffer_size;       /* mapper scan_buffer_size */
} kh_decay_params;
KH_API void kh_decay_params_default(kh_decay_params * p);
/* kept[k] = 0 for candidates computeScores erases (IoU below iou_thresh or fewer than 2 edges); scores[k] is
 * computeScore's return for the kept ones.  Output pointers may be NULL. */
KH_API int kh_lifelong_scores(int32_t device, const kh_scan_box * reference, int32_t n, const kh_scan_box * candidates,
                              const kh_decay_params * params, int32_t * kept, double * iou, double * area_overlap,
                              double * reading_overlap, double * scores);

/* ---------------------------------------------------------------- mapper front end (BASELINE configs 1 and 5) */
/* ROS-free restatement of what karto::Mapper::Process does around the scan matcher and the solver plugin
 * (lib/karto_sdk/src/Mapper.cpp:2679-2748 with MapperGraph::AddEdges / LinkNearChains / TryCloseLoop / CorrectPoses,
 * :1434-1561, 1641-1681, 2012-2030), for replaying a scan queue end to end on the GPU: sequential match against the running
 * scans, links to the previous scan, the running chain and the near chains (matched as one batch), loop closure as
 * speculative batches (all candidate chains enumerated by kh_graph_find_loop_candidates_from, coarse-matched in one
 * kh_matcher_match_batch, the ones passing the coarse gate fine-matched in a second, results consumed in the reference's
 * order up to the first accepted closure, then kh_spa_compute and re-enumeration behind it).  One laser, zero mount
 * offset, mapping mode.  The values are AS STORED by karto::Mapper (loop_match_maximum_variance_coarse and the two
 * variance penalties of `match` are the squared values). */
typedef struct kh_mapper kh_mapper;
typedef struct kh_laser {                       /* karto::LaserRangeFinder (Karto.h:4060-4330) */
  int32_t n_beams;
  double minimum_angle, angular_resolution, minimum_range, maximum_range, range_threshold;
  double offset_x, offset_y, offset_heading;    /* LaserRangeFinder::GetOffsetPose: where the sensor sits on the robot (zeros = at
                                                   its centre); sensor pose = GetSensorAt(corrected pose), Karto.h:5566-5588 */
} kh_laser;
typedef struct kh_mapper_params {               /* Mapper::InitializeParameters (Mapper.cpp:2086-2297) */
  int32_t use_scan_matching, use_scan_barycenter;
  double minimum_time_interval, minimum_travel_distance, minimum_travel_heading;
  int32_t scan_buffer_size;
  double scan_buffer_maximum_scan_distance;
  double link_match_minimum_response_fine, link_scan_maximum_distance, loop_search_maximum_distance;
  int32_t do_loop_closing, loop_match_minimum_chain_size;
  double loop_match_maximum_variance_coarse, loop_match_minimum_response_coarse, loop_match_minimum_response_fine;
  double correlation_search_space_dimension, correlation_search_space_resolution, correlation_search_space_smear_deviation;
  double loop_search_space_dimension, loop_search_space_resolution, loop_search_space_smear_deviation;
  kh_match_params match;
} kh_mapper_params;
typedef struct kh_mapper_stats {
  int64_t scans_processed, matches, loop_candidates, loop_closures, speculation_discarded, nodes_removed;
  double process_ms, match_ms, solver_ms, update_ms, lifelong_ms;
  int64_t fused_declined, fused_declined_reason;   /* ... that went the general way, and why the last one did (kh_matcher_seq_stats [6], [7]) */
  int64_t fused_matches, fused_fine_passes;     /* sequential matches that took the fused path of one MatchScan / whose fine pass
                                                   the device finished (kh_matcher_seq_stats of the sequential matcher) */
} kh_mapper_stats;
/* config/mapper_params_offline.yaml:31-66 */
KH_API void kh_mapper_params_default(kh_mapper_params * p);
/* max_candidates = capacity of one matcher batch (near chains / loop candidates beyond it go in further batches) */
KH_API int kh_mapper_create(const kh_mapper_params * params, const kh_laser * laser, int32_t device, int32_t max_candidates,
                            kh_mapper ** out);
/* The same mapper with its candidate batches (loop closure, near chains) dealt over one matcher pair per entry of `devices`
 * (kh_matcher_group); devices[0] also carries the sequential matches, the solver and the graph store.  The run is
 * identical to the one-device mapper's: the matches are independent and are consumed in candidate order. */
KH_API int kh_mapper_create_on_devices(const kh_mapper_params * params, const kh_laser * laser, const int32_t * devices,
                                       int32_t n_devices, int32_t max_candidates, kh_mapper ** out);
KH_API void kh_mapper_destroy(kh_mapper * m);
/* Mapper::Process for one scan: `ranges` = laser->n_beams readings, the odometric pose of the robot, the time stamp.
 * *accepted = 0 when the scan is dropped by HasMovedEnough (Mapper.cpp:3110-3142).  corrected_pose / covariance may be NULL. */
KH_API int kh_mapper_process(kh_mapper * m, const double * ranges, const double odometric_pose[3], double time,
                             int32_t * accepted, double corrected_pose[3], double covariance[9]);
KH_API int32_t kh_mapper_num_scans(const kh_mapper * m);
KH_API int64_t kh_mapper_num_edges(const kh_mapper * m);
KH_API int kh_mapper_get_poses(const kh_mapper * m, double * corrected_poses /* 3 * num_scans */);
/* scan `index` as the matcher / occupancy grid / lifelong scoring read it (pointers stay valid until the next process) */
KH_API int kh_mapper_get_scan(const kh_mapper * m, int32_t index, kh_scan * scan, kh_scan_box * box);
KH_API int kh_mapper_get_stats(const kh_mapper * m, kh_mapper_stats * out);
/* Mapper::RemoveNodeFromGraph + MapperSensorManager::RemoveScan (Mapper.cpp:2964-3021, :208-218; what
 * LifelongSlamToolbox::removeFromSlamGraph does, slam_toolbox_lifelong.cpp:330-342): the scan's edges leave its
 * neighbours, the graph and the solver (RemoveConstraint), the node leaves the solver (RemoveNode) and the scan list. */
KH_API int kh_mapper_remove_node(kh_mapper * m, int32_t scan_id);
/* Vertex::GetAdjacentVertices of the scan's vertex, in the reference's order (Mapper.h:338-361): *n = their number,
 * `adjacent` (may be NULL) receives at most `capacity` scan ids.  Vertex::SetScore (Mapper.h:326-329), what
 * LifelongSlamToolbox::updateScoresSlamGraph does to a vertex that stays (slam_toolbox_lifelong.cpp:356-365).  Together
 * with kh_mapper_get_scan and kh_mapper_remove_node a host can run its own node-decay policy -- and the tests replay the
 * library's policy with an independent restatement. */
KH_API int kh_mapper_get_adjacency(const kh_mapper * m, int32_t scan_id, int32_t * adjacent, int32_t capacity, int32_t * n);
KH_API int kh_mapper_set_node_score(kh_mapper * m, int32_t scan_id, double score);
/* LifelongSlamToolbox::evaluateNodeDepreciation after every accepted scan (slam_toolbox_lifelong.cpp:149-178):
 * FindNearLinkedVertices within half the diagonal of the scan's bounding box, kh_lifelong_scores over them, removal of the
 * ones scoring below params->removal_score, the new score stored on the others.  params = NULL switches it off. */
KH_API int kh_mapper_set_lifelong(kh_mapper * m, const kh_decay_params * params);
/* ids of the scans still in the graph, ascending (ids[kh_mapper_num_alive]) */
KH_API int32_t kh_mapper_num_alive(const kh_mapper * m);
KH_API int kh_mapper_get_alive(const kh_mapper * m, int32_t * ids);
/* the solver plugin instance the mapper drives (RemoveNode / save / load ... ); owned by the mapper */
KH_API kh_spa * kh_mapper_solver(kh_mapper * m);
/* every solver call the mapper makes, one line each, in the format oracle/ref_slam_driver.cpp logs the reference
 * Mapper's calls with (N id pose, C a b z cov, X n ms, P id pose, K): the two logs of one scan queue must agree */
KH_API int kh_mapper_set_log(kh_mapper * m, const char * path);

#ifdef __cplusplus
}
#endif
#endif  /* KARTO_HIP_H_ */
