"""GPU parity of the HIP scan matcher (through the C ABI) against the CPU oracle on identical seeded
inputs.  Bar: grid bytes, lookup-table indices and integer response sums bit-exact; responses, best
pose and covariance bit-identical doubles (the host half of the product repeats the reference's
IEEE operation order, the device half is exact integer + unfused FP64)."""
import numpy as np
import pytest

from common import C2_PARAMS, PRESETS, Scenario, bits, make_hip_matcher, make_oracle_matcher

pytestmark = pytest.mark.gpu


def _assert_same(a, b, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(bits(a), bits(b)), f"{what}: {a} vs {b} (max abs diff {np.max(np.abs(a - b))})"


@pytest.mark.parametrize("preset", ["K", "S", "L"])
def test_match_scan_parity(kartohip_lib, preset):
    sc = Scenario(seed=11, n_base=10, start=20)
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher(preset)
    hm = make_hip_matcher(preset)
    assert np.array_equal(om.kernel(), hm.kernel())
    for pen, refine in [(True, True), (False, True), (False, False)]:
        r_o, mean_o, cov_o = om.match_scan(oq, ob, pen, refine)
        r_h, mean_h, cov_h = hm.MatchScan(hq, hb, pen, refine)
        assert np.array_equal(om.grid(), hm.GetCorrelationGrid()), "rasterised grid differs"
        assert np.array_equal(om.lookup_table(), hm.lookup_table()), "lookup table differs"
        _assert_same(r_o, r_h, "response")
        _assert_same(mean_o, mean_h, "mean")
        _assert_same(cov_o, cov_h, "covariance")
    hm.close()


@pytest.mark.parametrize("preset,fine", [("K", False), ("L", False), ("S", True)])
def test_correlate_volume_parity(kartohip_lib, preset, fine):
    """Every pose of the response volume, not just the winner."""
    sc = Scenario(seed=5, n_base=8, start=60, perturb=(-0.04, 0.06, -0.03))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher(preset)
    hm = make_hip_matcher(preset)
    hm.set_debug(True)
    om.add_scans(oq, ob)
    hm.AddScans(hq, hb)
    assert np.array_equal(om.grid(), hm.GetCorrelationGrid())
    res = 1.0 / om.grid_info()["scale"]
    p = PRESETS[preset]["params"]
    if fine:
        args = ((res, res), (res, res), 0.5 * p["coarse_angle_resolution"], p["fine_search_angle_offset"])
    else:
        side = PRESETS[preset]["create"][0]
        off = 0.5 * round(side / res) * res
        args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
    for pen in (True, False):
        r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, pen, fine)
        r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, pen, None, fine)
        assert np.array_equal(om.lookup_table(), hm.lookup_table())
        vol = om.volume()          # (ny, nx, na, 4)
        sums, resp = hm.volume()
        assert resp.shape == vol.shape[:3]
        assert np.array_equal(bits(vol[..., 0]), bits(resp)), "response volume differs"
        _assert_same(r_o, r_h, "response")
        _assert_same(mean_o, mean_h, "mean")
        _assert_same(cov_o, cov_h, "covariance")
    hm.close()


def test_config2_correlate(kartohip_lib):
    """BASELINE config 2: 61 x 61 x 81 poses x 1081 beams on the 8087^2 grid."""
    import math
    sc = Scenario(seed=7, n_base=10, start=0)
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher("C2", threads=8)
    hm = make_hip_matcher("C2")
    hm.set_debug(True)
    om.add_scans(oq, ob)
    hm.AddScans(hq, hb)
    assert np.array_equal(om.grid(), hm.GetCorrelationGrid())
    args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
    r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, False)
    r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, False)
    assert np.array_equal(om.lookup_table(), hm.lookup_table())
    vol = om.volume()
    sums, resp = hm.volume()
    assert vol.shape[:3] == (61, 61, 81)
    assert np.array_equal(bits(vol[..., 0]), bits(resp))
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    _assert_same(cov_o, cov_h, "covariance")
    hm.close()


def test_batch_equals_single(kartohip_lib):
    """A batch of independent matches returns exactly what one-at-a-time calls return."""
    scs = [Scenario(seed=20 + i, n_base=6 + i, start=30 * i + 5, perturb=(0.03 * i, -0.02, 0.01 * i)) for i in range(4)]
    hm1 = make_hip_matcher("K")
    hmb = make_hip_matcher("K", max_batch=4)
    singles = []
    qs, bs = [], []
    for sc in scs:
        q, b = sc.hip_scans()
        qs.append(q)
        bs.append(b)
        singles.append(hm1.MatchScan(q, b))
    resp, means, covs, status = hmb.MatchScanBatch(qs, bs)
    assert (status == 0).all()
    for i, (r, m, c) in enumerate(singles):
        _assert_same(r, resp[i], "response")
        _assert_same(m, means[i], "mean")
        _assert_same(c, covs[i], "cov")
    hm1.close()
    hmb.close()


def test_chunked_batch_equals_single(kartohip_lib):
    """Batches of >= 128 matches go through the library in chunks of 64 on two staging sets, the host half of one
    chunk overlapping the kernels of another (matcher_host.cpp::correlate_batch).  150 matches (8 distinct pairs,
    tiled; the last chunk is partial) must come back exactly as one-at-a-time calls return them."""
    scs = [Scenario(seed=60 + i, n_base=5 + i % 4, start=23 * i + 2, perturb=(0.02 * i, -0.01 * i, 0.005 * i)) for i in range(8)]
    hm1 = make_hip_matcher("K")
    singles, pairs = [], []
    for sc in scs:
        q, b = sc.hip_scans()
        pairs.append((q, b))
        singles.append(hm1.MatchScan(q, b))
    hm1.close()
    n = 150
    hmb = make_hip_matcher("K", max_batch=n)
    hmb.set_debug(False, force_chunks=True)              # small searches are not chunked on their own
    qs = [pairs[i % 8][0] for i in range(n)]
    bs = [pairs[i % 8][1] for i in range(n)]
    for _ in range(2):                                   # second pass: staging sets and slots reused
        resp, means, covs, status = hmb.MatchScanBatch(qs, bs)
        assert (status == 0).all()
        for i in range(n):
            r, m, c = singles[i % 8]
            _assert_same(r, resp[i], "response")
            _assert_same(m, means[i], "mean")
            _assert_same(c, covs[i], "cov")
    hmb.close()


def test_chunked_config2_batch_equals_single(kartohip_lib):
    """The bench path: 130 config-2 CorrelateScans in one call (chunked by the library on its own: 64 + 64 + 2)
    against the same searches one at a time, rasterised grids resident in their slots."""
    import math
    from slam_toolbox_amd.scan_matcher import _scan_array
    n = 130
    scs = [Scenario(seed=70 + i, n_base=6, start=41 * i + 3, perturb=(0.03 * (i - 1), 0.02 * i, 0.01 * (i - 2))) for i in range(4)]
    hm = make_hip_matcher("C2", max_batch=n)
    queries, centers = [], []
    for b in range(n):
        q, base = scs[b % 4].hip_scans()
        hm.AddScans(q, base, slot=b)
        queries.append(q)
        centers.append(scs[b % 4].query_pose)
    corr = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
    singles = [hm.CorrelateScan(queries[b], centers[b], *corr, True, None, False, slot=b) for b in range(4)]
    resp, means, covs, status = hm.CorrelateScanBatch(None, np.asarray(centers), *corr, True, False,
                                                      scan_array=(_scan_array(queries), n))
    assert (status == 0).all()
    for b in range(n):
        r, m, c = singles[b % 4]
        _assert_same(r, resp[b], "response")
        _assert_same(m, means[b], "mean")
        _assert_same(c, covs[b], "cov")
    hm.close()


def test_chunked_batch_of_searches_three_tiles_tall(kartohip_lib):
    """130 one-cell searches of 81 x 81 poses (three scoring tiles of 28 rows) in one call -- the chunked pipeline with
    a kernel instance and a tile count config 2 never uses -- against the same searches one at a time."""
    import math
    from slam_toolbox_amd.scan_matcher import MapperParams, ScanMatcher, _scan_array
    n = 130
    scs = [Scenario(seed=90 + i, n_base=6, start=37 * i + 5, perturb=(0.02 * (i - 1), 0.03 * i, 0.01 * (i - 2))) for i in range(4)]
    hm = ScanMatcher.Create(MapperParams(**C2_PARAMS), 0.4, 0.005, 0.03, 20.0, max_batch=n)
    queries, centers = [], []
    for b in range(n):
        q, base = scs[b % 4].hip_scans()
        hm.AddScans(q, base, slot=b)
        queries.append(q)
        centers.append(scs[b % 4].query_pose)
    corr = ((0.2, 0.2), (0.005, 0.005), math.radians(10.0), math.radians(0.5))
    singles = [hm.CorrelateScan(queries[b], centers[b], *corr, True, None, False, slot=b) for b in range(4)]
    resp, means, covs, status = hm.CorrelateScanBatch(None, np.asarray(centers), *corr, True, False,
                                                      scan_array=(_scan_array(queries), n))
    assert (status == 0).all()
    for b in range(n):
        r, m, c = singles[b % 4]
        _assert_same(r, resp[b], "response")
        _assert_same(m, means[b], "mean")
        _assert_same(c, covs[b], "cov")
    hm.close()


def test_empty_grid_and_empty_scan(kartohip_lib):
    """No base scans -> every pose ties at response 0 (host fallback path); empty query scan -> the
    reference's early return (Mapper.cpp:547-557)."""
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    from oracle import karto
    from common import LASER
    sc = Scenario(seed=3, n_base=1, start=10)
    oq, _ = sc.oracle_scans()
    hq, _ = sc.hip_scans()
    om = make_oracle_matcher("K")
    hm = make_hip_matcher("K")
    r_o, mean_o, cov_o = om.match_scan(oq, [], True, True)
    r_h, mean_h, cov_h = hm.MatchScan(hq, [], True, True)
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    _assert_same(cov_o, cov_h, "cov")
    empty_o = karto.Scan(np.zeros(0), sc.query_pose, LASER, points=np.zeros((0, 2)))
    empty_h = LocalizedRangeScan(np.zeros(0), sc.query_pose, LASER.min_angle, LASER.ang_res)
    r_o, mean_o, cov_o = om.match_scan(empty_o, [], True, True)
    r_h, mean_h, cov_h = hm.MatchScan(empty_h, [], True, True)
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    _assert_same(cov_o, cov_h, "cov")
    hm.close()


def test_response_expansion_path(kartohip_lib):
    """offline.yaml has use_response_expansion: a zero coarse response widens the angular search by 20 degrees up
    to three times on the same grid (Mapper.cpp:594-619).  An empty grid forces all three rounds."""
    sc = Scenario(seed=9, n_base=1, start=40)
    oq, _ = sc.oracle_scans()
    hq, _ = sc.hip_scans()
    om = make_oracle_matcher("L")
    hm = make_hip_matcher("L")
    r_o, mean_o, cov_o = om.match_scan(oq, [], False, True)
    r_h, mean_h, cov_h = hm.MatchScan(hq, [], False, True)
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    _assert_same(cov_o, cov_h, "cov")
    hm.close()


@pytest.mark.parametrize("search_res", [0.015, 0.03, 0.007])
def test_search_resolution_off_the_grid_pitch(kartohip_lib, search_res):
    """CorrelateScan is public and takes any search resolution: when it is not 1x or 2x the grid pitch the
    lattice's grid indices are no longer an arithmetic progression (or step by 3 cells) and the kernel's
    per-pose exact path runs instead of the windowed one.  Every pose of the volume must still match."""
    sc = Scenario(seed=6, n_base=6, start=90, perturb=(0.02, 0.03, 0.02))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher("K")
    hm = make_hip_matcher("K")
    hm.set_debug(True)
    om.add_scans(oq, ob)
    hm.AddScans(hq, hb)
    off = 6 * search_res
    args = ((off, off), (search_res, search_res), 0.1, 0.02)
    r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, True)
    r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, True)
    vol = om.volume()
    sums, resp = hm.volume()
    assert resp.shape == vol.shape[:3]
    assert np.array_equal(bits(vol[..., 0]), bits(resp)), "response volume differs"
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    hm.close()


@pytest.mark.parametrize("preset,fine", [("K", False), ("S", False), ("S", True), ("C2", False)])
def test_lds_staged_scoring_path(kartohip_lib, preset, fine):
    """The LDS-staged scoring kernels (k_offsets_lds / k_score_lds: the union of the windows of a run of consecutive beams
    at two adjacent angles staged through LDS, byte sums on the matrix cores) must produce the same volume bit for bit -- on
    every search shape they can take, not only the large ones they score by default; K also exercises the fallback of beams
    whose windows at the two angles are too far apart for one region."""
    import math
    sc = Scenario(seed=5, n_base=8, start=60, perturb=(-0.04, 0.06, -0.03))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher(preset, threads=8)
    hm = make_hip_matcher(preset)
    hm.set_debug(True, lds_score=True)
    om.add_scans(oq, ob)
    hm.AddScans(hq, hb)
    res = 1.0 / om.grid_info()["scale"]
    p = PRESETS[preset]["params"]
    if preset == "C2":
        args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
    elif fine:
        args = ((res, res), (res, res), 0.5 * p["coarse_angle_resolution"], p["fine_search_angle_offset"])
    else:
        side = PRESETS[preset]["create"][0]
        off = 0.5 * round(side / res) * res
        args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
    r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, fine)
    r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, fine)
    vol = om.volume()
    sums, resp = hm.volume()
    assert np.array_equal(om.lookup_table(), hm.lookup_table())
    assert np.array_equal(bits(vol[..., 0]), bits(resp)), "response volume differs"
    _assert_same(r_o, r_h, "response")
    _assert_same(mean_o, mean_h, "mean")
    hm.close()


@pytest.mark.parametrize("preset", ["S", "L", "C2"])
def test_empty_window_skipping_is_invisible(kartohip_lib, preset):
    """K2 leaves out the beams whose whole search window lies in 32 x 32 grid blocks no scan point was stamped
    into (they add 0 to every pose).  Default (skipping) and dense scoring (kh_matcher_set_debug bit 2) must give
    the same integer sums, both equal to the oracle's volume; a query far away from every base scan (all windows
    empty) must come back with response 0 exactly like the oracle."""
    import math
    sc = Scenario(seed=15, n_base=9, start=40, perturb=(0.03, -0.05, 0.04))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher(preset, threads=8)
    res = None
    vols = []
    for dense, windowed in ((False, False), (True, False), (False, True), (True, True)):
        hm = make_hip_matcher(preset)
        hm.set_debug(True, dense_score=dense, windowed_score=windowed)
        hm.AddScans(hq, hb)
        if res is None:
            om.add_scans(oq, ob)
            res = 1.0 / om.grid_info()["scale"]
            p = PRESETS[preset]["params"]
            if preset == "C2":
                args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
            else:
                side = PRESETS[preset]["create"][0]
                off = 0.5 * round(side / res) * res
                args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
            r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, False)
            vol = om.volume()
        r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, False)
        sums, resp = hm.volume()
        assert np.array_equal(bits(vol[..., 0]), bits(resp)), "response volume differs"
        _assert_same(r_o, r_h, "response")
        _assert_same(mean_o, mean_h, "mean")
        _assert_same(cov_o, cov_h, "covariance")
        vols.append(sums)
        # a search centred 15 m away from the map: nothing but zeros in every window
        far = sc.query_pose + np.array([15.0, 15.0, 0.3])
        r_of, mean_of, cov_of = om.correlate_scan(oq, far, *args, True, False)
        r_hf, mean_hf, cov_hf = hm.CorrelateScan(hq, far, *args, True, None, False)
        _assert_same(r_of, r_hf, "far response")
        _assert_same(mean_of, mean_hf, "far mean")
        _assert_same(cov_of, cov_hf, "far covariance")
        hm.close()
    assert np.array_equal(vols[0], vols[1])
    assert vols[0].max() > 0


def test_ragged_batch_and_all_invalid_scan(kartohip_lib):
    """One batch mixing scans of different beam counts and chain lengths, plus a query whose ranges are all NaN /
    inf (every table entry INVALID_SCAN, response 0 everywhere -> the tie fallback): each result must equal the
    oracle's for that pair alone."""
    from common import LASER
    from oracle import karto
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    om = make_oracle_matcher("K")
    hmb = make_hip_matcher("K", max_batch=4)
    qs, bs, expect = [], [], []
    for i, (n_base, keep) in enumerate([(3, 1081), (12, 541), (7, 1081), (5, 700)]):
        sc = Scenario(seed=40 + i, n_base=n_base, start=25 * i + 3, perturb=(0.02 * i, 0.01, -0.01 * i))
        ranges_q = sc.query_ranges[:keep].copy()
        if i == 2:
            ranges_q[::2] = np.nan
            ranges_q[1::2] = np.inf
        oq = karto.Scan(ranges_q, sc.query_pose, LASER)
        ob = [karto.Scan(sc.ranges[k][:keep], sc.base_poses[k], LASER) for k in range(n_base)]
        qs.append(LocalizedRangeScan(ranges_q, sc.query_pose, LASER.min_angle, LASER.ang_res))
        bs.append([LocalizedRangeScan(sc.ranges[k][:keep], sc.base_poses[k], LASER.min_angle, LASER.ang_res) for k in range(n_base)])
        expect.append(om.match_scan(oq, ob, True, True))
    resp, means, covs, status = hmb.MatchScanBatch(qs, bs)
    assert (status == 0).all()
    for i, (r, m, c) in enumerate(expect):
        _assert_same(r, resp[i], f"response {i}")
        _assert_same(m, means[i], f"mean {i}")
        _assert_same(c, covs[i], f"cov {i}")
    assert resp[2] == 0.0
    hmb.close()


def test_dual_copy_layout_follows_the_grid(kartohip_lib):
    """The re-pitched copies the full-resolution search reads (CorrJob::grid2) are built at the first search and then kept in
    step with the grid tile by tile: a second, different scene rasterised into the same slot must score exactly like the
    oracle again (stale tiles zeroed, new ones copied), and switching the copies off must not change a bit."""
    import math
    args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
    om = make_oracle_matcher("C2", threads=8)
    hm = make_hip_matcher("C2")
    hm.set_debug(True, windowed_score=True)             # the windowed kernel (a config-2 search takes the LDS-staged one by default)
    results = []
    for seed, start in ((7, 0), (8, 120), (9, 40)):
        sc = Scenario(seed=seed, n_base=10, start=start)
        oq, ob = sc.oracle_scans()
        hq, hb = sc.hip_scans()
        om.add_scans(oq, ob)
        hm.AddScans(hq, hb)
        r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, False)
        r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, False)
        vol = om.volume()
        sums, resp = hm.volume()
        assert np.array_equal(bits(vol[..., 0]), bits(resp)), f"scene {seed}: response volume differs"
        _assert_same(r_o, r_h, "response"); _assert_same(mean_o, mean_h, "mean"); _assert_same(cov_o, cov_h, "covariance")
        results.append((hq, sc.query_pose, sums.copy()))
    hm.set_debug(True, no_dual_copy=True, windowed_score=True)
    hq, pose, sums = results[-1]
    hm.CorrelateScan(hq, pose, *args, True, None, False)
    sums2, _ = hm.volume()
    assert np.array_equal(sums, sums2)
    # the matrix-core instance of the scoring kernel (kh_matcher_set_debug bit 5): the same integer sums
    for no_copies in (False, True):
        hm.set_debug(True, no_dual_copy=no_copies, mfma_score=True, windowed_score=True)
        hm.CorrelateScan(hq, pose, *args, True, None, False)
        sums3, _ = hm.volume()
        assert np.array_equal(sums, sums3), f"MFMA scoring differs (copies off: {no_copies})"
    # ... and the LDS-staged kernels (what a batch of these searches takes by default), same slot, same grid
    hm.set_debug(True, lds_score=True)
    hm.CorrelateScan(hq, pose, *args, True, None, False)
    sums4, _ = hm.volume()
    assert np.array_equal(sums, sums4), "LDS-staged scoring differs from the windowed kernel"
    hm.close()


def test_two_cell_search_from_the_column_decimated_copies(kartohip_lib):
    """MatchScan's coarse pass of the loop preset steps two cells: the slot gets column-decimated copies (CorrJob::dec) and the
    search is scored as a one-cell search on them -- windows that start in the zero rows in front of the grid, run over a
    row end or leave the array included (readings up to 30 m on a grid with a 20 m border).  The sums must equal the
    oracle's volume, with the copies, without them, and through the matrix-core instance of the kernel."""
    import math
    sc = Scenario(seed=21, n_base=12, start=300, perturb=(0.9, -1.3, 0.05))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher("L", threads=8)
    om.add_scans(oq, ob)
    res = 1.0 / om.grid_info()["scale"]
    side = PRESETS["L"]["create"][0]
    off = 0.5 * round(side / res) * res
    p = PRESETS["L"]["params"]
    args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
    r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, False, False)
    vol = om.volume()
    for kw in (dict(), dict(no_dual_copy=True), dict(mfma_score=True)):
        hm = make_hip_matcher("L")
        hm.set_debug(True, **kw)
        hm.AddScans(hq, hb)
        r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, False, None, False)
        sums, resp = hm.volume()
        assert np.array_equal(bits(vol[..., 0]), bits(resp)), f"response volume differs ({kw})"
        _assert_same(r_o, r_h, "response"); _assert_same(mean_o, mean_h, "mean"); _assert_same(cov_o, cov_h, "covariance")
        hm.close()


@pytest.mark.parametrize("side_m, n_angles", [(0.4, 17), (0.64, 9)])
def test_full_resolution_search_of_several_tiles(kartohip_lib, side_m, n_angles):
    """One-cell searches taller than two scoring tiles: 81 x 81 poses (rows per lane 7: three tiles of 28 rows, one list for all
    tiles, copies A / B) and 129 x 129 poses (3 x 5 tiles of 61 x 32 poses with a list per tile).  config 2 and the presets
    never produce these shapes; the sums must still be the oracle's, with and without the copies."""
    import math
    from oracle import karto
    from slam_toolbox_amd.scan_matcher import MapperParams, ScanMatcher
    create = (side_m, 0.005, 0.03, 20.0)
    sc = Scenario(seed=33, n_base=8, start=200, perturb=(0.02, -0.04, 0.01))
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = karto.Matcher(*create, C2_PARAMS, threads=16)
    om.add_scans(oq, ob)
    off = 0.5 * round(side_m / 0.005) * 0.005
    half = 0.5 * (n_angles - 1) * math.radians(0.5)
    args = ((off, off), (0.005, 0.005), half, math.radians(0.5))
    r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, True, False)
    vol = om.volume()
    for kw in (dict(), dict(no_dual_copy=True), dict(mfma_score=True)):
        hm = ScanMatcher.Create(MapperParams(**C2_PARAMS), *create, max_batch=1)
        hm.set_debug(True, **kw)
        hm.AddScans(hq, hb)
        r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, True, None, False)
        sums, resp = hm.volume()
        assert resp.shape == vol[..., 0].shape
        assert np.array_equal(bits(vol[..., 0]), bits(resp)), f"response volume differs ({kw})"
        _assert_same(r_o, r_h, "response"); _assert_same(mean_o, mean_h, "mean"); _assert_same(cov_o, cov_h, "covariance")
        hm.close()


@pytest.mark.parametrize("preset", ["K", "S"])
def test_covariance_members_reproduce_the_search(kartohip_lib, preset):
    """ComputePositionalCovariance / ComputeAngularCovariance as separate calls (the two public members of
    karto::ScanMatcher nobody calls from outside, Mapper.h:1405-1427) give what CorrelateScan computed with them inside,
    which the tests above pin to the oracle: position block from the coarse search, theta-theta from the fine one."""
    sc = Scenario(seed=5, n_base=8, start=60, perturb=(-0.04, 0.06, -0.03))
    hq, hb = sc.hip_scans()
    hm = make_hip_matcher(preset)
    hm.AddScans(hq, hb)
    res = 1.0 / hm.grid_info()["scale"]
    p = PRESETS[preset]["params"]
    side = PRESETS[preset]["create"][0]
    off = 0.5 * round(side / res) * res
    coarse = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
    r, mean, cov = hm.CorrelateScan(hq, sc.query_pose, *coarse, True, None, False)
    again = hm.ComputePositionalCovariance(mean, r, sc.query_pose, coarse[0], coarse[1], coarse[3])
    _assert_same(cov, again, "positional covariance")
    fine = ((res, res), (res, res), 0.5 * p["coarse_angle_resolution"], p["fine_search_angle_offset"])
    r2, mean2, cov2 = hm.CorrelateScan(hq, mean, *fine, True, cov, True)
    tt = hm.ComputeAngularCovariance(hq, mean2, r2, mean, fine[2], fine[3])
    _assert_same(cov2[2, 2], tt, "angular covariance")
    hm.close()


def test_resident_base_scans_equal_uploaded_ones(kartohip_lib):
    """kh_scan.device_points_xy: a base scan kept in HBM is read where it lies -- same grid, same match as when the call
    uploads it; a pose change re-uploads (SetSensorPose), a chain may mix both kinds"""
    def flat(res):
        r, mean, cov = res
        return np.concatenate([[r], np.asarray(mean).reshape(3), np.asarray(cov).reshape(9)])
    sc = Scenario(seed=21, n_base=12, start=40)
    q, base = sc.hip_scans()
    hm = make_hip_matcher("S")
    want = flat(hm.MatchScan(q, base, True, True))
    grid = hm.GetCorrelationGrid().copy()
    for k, b in enumerate(base):
        if k % 3 != 1:
            b.MakeResident()
    got = flat(hm.MatchScan(q, base, True, True))
    assert np.array_equal(hm.GetCorrelationGrid(), grid)
    assert np.array_equal(bits(got), bits(want))
    # move one resident scan: the device copy must follow
    base[0].SetSensorPose(base[0].GetSensorPose() + np.array([0.07, -0.04, 0.01]))
    got2 = flat(hm.MatchScan(q, base, True, True))
    fresh = [type(b)(b.ranges, b.GetSensorPose(), b.min_angle, b.angular_resolution) for b in base]
    want2 = flat(hm.MatchScan(q, fresh, True, True))
    assert np.array_equal(bits(got2), bits(want2)) and not np.array_equal(bits(got2), bits(want))
    hm.close()
