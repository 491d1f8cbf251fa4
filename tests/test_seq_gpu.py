"""The fused path of ONE MatchScan (csrc/matcher_seq.cpp, kh_matcher_match -- the only call the reference's API makes,
Mapper.cpp:534-639, 2714-2717) against the general path of the same library (kh_matcher_set_debug bit 7) and the CPU oracle:
response, mean, covariance, rasterised grid, lookup table and the stored volume of the last search, bit for bit, on every
preset, with and without penalties / refinement, base scans uploaded and resident, and the cases the device hands back to the
general path (empty grid, several best poses, response expansion)."""
import numpy as np
import pytest

from common import PRESETS, Scenario, bits, make_hip_matcher, make_oracle_matcher

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert np.array_equal(bits(a), bits(b)), f"{what}: {a} vs {b}"


def _pair(preset):
    fused = make_hip_matcher(preset)
    general = make_hip_matcher(preset)
    general.set_debug(False, no_fused_match=True)
    return fused, general


@pytest.mark.parametrize("preset", ["K", "S", "L", "C2"])
@pytest.mark.parametrize("seed,n_base,start,perturb", [(11, 10, 20, (0.05, -0.03, 0.02)), (3, 20, 150, (-0.02, 0.04, -0.015)),
                                                        (29, 5, 300, (0.0, 0.0, 0.0))])
def test_fused_equals_general_and_oracle(kartohip_lib, preset, seed, n_base, start, perturb):
    sc = Scenario(seed=seed, n_base=n_base, start=start, perturb=perturb)
    oq, ob = sc.oracle_scans()
    hq, hb = sc.hip_scans()
    om = make_oracle_matcher(preset, threads=8 if preset == "C2" else 1)
    fused, general = _pair(preset)
    for pen, refine in [(True, True), (False, True), (True, False)]:
        r_o, mean_o, cov_o = om.match_scan(oq, ob, pen, refine)
        r_f, mean_f, cov_f = fused.MatchScan(hq, hb, pen, refine)
        r_g, mean_g, cov_g = general.MatchScan(hq, hb, pen, refine)
        for what, o, f, g in (("response", r_o, r_f, r_g), ("mean", mean_o, mean_f, mean_g), ("covariance", cov_o, cov_f, cov_g)):
            _same(g, f, f"{what} (fused vs general, pen={pen} refine={refine})")
            _same(o, f, f"{what} (fused vs oracle, pen={pen} refine={refine})")
        assert np.array_equal(om.grid(), fused.GetCorrelationGrid()), "rasterised grid differs from the oracle's"
        assert np.array_equal(general.GetCorrelationGrid(), fused.GetCorrelationGrid())
        assert np.array_equal(om.lookup_table(), fused.lookup_table()), "lookup table of the last search differs"
        sf, _ = fused.volume(responses=False)
        sg, _ = general.volume(responses=False)
        assert np.array_equal(sf, sg), "stored sums of the last search differ"
    st = fused.seq_stats()
    assert st["calls"] == 3 and general.seq_stats()["calls"] == 0
    assert st["fine_mismatches"] == 0
    assert st["fine_on_device"] + st["fine_fallbacks"] == 2
    assert st["fused_score"] == 3                # every linear lattice: table + scoring in one launch
    fused.close(); general.close()


@pytest.mark.parametrize("preset", ["S", "K"])
def test_fused_with_resident_base_scans_and_repeated_calls(kartohip_lib, preset):
    """What kh_mapper does: base scans resident in HBM, one matcher, call after call with changing chains (the first-point table,
    the previous-tiles list and the result flag must be left clean by every call)."""
    scs = [Scenario(seed=40 + i, n_base=4 + 3 * i, start=37 * i + 3, perturb=(0.01 * i, -0.02, 0.004 * i)) for i in range(5)]
    fused, general = _pair(preset)
    om = make_oracle_matcher(preset)
    for rnd in range(2):
        for sc in scs:
            hq, hb = sc.hip_scans()
            if rnd == 1:
                for b in hb:
                    b.MakeResident(0)
            r_f, mean_f, cov_f = fused.MatchScan(hq, hb, True, True)
            r_g, mean_g, cov_g = general.MatchScan(hq, hb, True, True)
            _same(r_g, r_f, "response"); _same(mean_g, mean_f, "mean"); _same(cov_g, cov_f, "covariance")
            assert np.array_equal(general.GetCorrelationGrid(), fused.GetCorrelationGrid())
            oq, ob = sc.oracle_scans()
            r_o, mean_o, cov_o = om.match_scan(oq, ob, True, True)
            _same(r_o, r_f, "response vs oracle"); _same(mean_o, mean_f, "mean vs oracle"); _same(cov_o, cov_f, "covariance vs oracle")
    st = fused.seq_stats()
    assert st["calls"] == 10 and st["fine_mismatches"] == 0 and st["fine_on_device"] >= 6
    fused.close(); general.close()


def test_fused_hands_degenerate_searches_back(kartohip_lib):
    """Nothing rasterised (base scans far away): every pose ties at 0 -- more ties than the result block holds; with response
    expansion the coarse pass is repeated by the general path.  Results equal the general path's."""
    sc = Scenario(seed=11, n_base=6, start=20)
    hq, hb = sc.hip_scans()
    far = hq.GetSensorPose() + np.array([500.0, 500.0, 0.0])
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    from common import LASER
    q_far = LocalizedRangeScan(sc.query_ranges, far, LASER.min_angle, LASER.ang_res)
    for preset in ("S", "K"):
        fused, general = _pair(preset)
        r_f, mean_f, cov_f = fused.MatchScan(q_far, hb, True, True)
        r_g, mean_g, cov_g = general.MatchScan(q_far, hb, True, True)
        _same(r_g, r_f, "response"); _same(mean_g, mean_f, "mean"); _same(cov_g, cov_f, "covariance")
        assert r_f == 0.0
        st = fused.seq_stats()
        assert st["calls"] == 1 and st["coarse_fallbacks"] == 1 and st["fine_on_device"] == 0
        # and the handle goes on working
        r_f, mean_f, cov_f = fused.MatchScan(hq, hb, True, True)
        r_g, mean_g, cov_g = general.MatchScan(hq, hb, True, True)
        _same(r_g, r_f, "response"); _same(mean_g, mean_f, "mean"); _same(cov_g, cov_f, "covariance")
        fused.close(); general.close()


def test_fused_ragged_and_null_scans(kartohip_lib):
    """Base scans of different lengths, NaN runs and an empty scan in the chain (Mapper.cpp:1039-1041, 1127-1136)."""
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    from common import LASER
    sc = Scenario(seed=17, n_base=7, start=90)
    hq, hb = sc.hip_scans()
    r2 = np.array(sc.ranges[2], dtype=np.float64); r2[100:180] = np.nan; r2[700] = np.inf
    hb[2] = LocalizedRangeScan(r2, sc.base_poses[2], LASER.min_angle, LASER.ang_res)
    hb[4] = LocalizedRangeScan(np.array(sc.ranges[4][:600]), sc.base_poses[4], LASER.min_angle, LASER.ang_res)
    hb[5] = LocalizedRangeScan(np.zeros(0), sc.base_poses[5], LASER.min_angle, LASER.ang_res)
    for preset in ("S", "K", "L"):
        fused, general = _pair(preset)
        r_f, mean_f, cov_f = fused.MatchScan(hq, hb, True, True)
        r_g, mean_g, cov_g = general.MatchScan(hq, hb, True, True)
        _same(r_g, r_f, "response"); _same(mean_g, mean_f, "mean"); _same(cov_g, cov_f, "covariance")
        assert np.array_equal(general.GetCorrelationGrid(), fused.GetCorrelationGrid())
        assert fused.seq_stats()["calls"] == 1
        fused.close(); general.close()
