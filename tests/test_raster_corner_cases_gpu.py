"""GPU: the corner cases of the rasteriser's data-parallel formulations pushed THROUGH THE KERNELS (kh_matcher_add_scans ->
kh_matcher_read_grid), not through numpy twins:

  * FindValidPoints (K0, k_find_valid_par: pointer doubling in LDS; k_find_valid_lane for scans of more than 2048 readings,
    csrc/matcher_kernels.hip): NaN runs, +inf beams, scans where every reading is a trigger, dozens of readings between
    triggers, leading NaNs with the viewpoint away from the sensor, all-NaN scans, a single valid reading, ragged short
    scans, jagged near / far returns, and 4000-reading scans -- the inputs of tests/test_find_valid_algorithm.py;
  * the order-dependent "cell already 100" rule (K1a, k_cell_first / k_cell_links / k_active_set: first point per cell +
    earlier-neighbour fixpoint): dense walls revisited by later scans, a filled block, a long staircase (a dependency chain
    as long as the staircase) and their reversals, at sigma / resolution = 10 (41 x 41 stamps) and at the loop preset --
    the inputs of tests/test_active_set_algorithm.py turned into point readings.

Every grid must equal, byte for byte, the grid of oracle/karto_oracle.c (pinned to the reference build, oracle/_ref, by
tests/test_oracle_vs_reference.py); the range-built cases are compared with the reference's own ScanMatcher as well when
oracle/_ref is there (it travels to the GPU box)."""
import numpy as np
import pytest

from common import LASER, OFFLINE_PARAMS, PRESETS, make_hip_matcher, make_oracle_matcher
from test_active_set_algorithm import _walls
from test_find_valid_algorithm import _cases, _range_cases

pytestmark = pytest.mark.gpu


def _pair(points, pose):
    """the same reading list as an oracle scan and as a library scan"""
    from oracle import karto
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    points = np.ascontiguousarray(points, dtype=np.float64)
    n = points.shape[0]
    o = karto.Scan(np.ones(n), np.asarray(pose, dtype=np.float64), points=points)
    h = LocalizedRangeScan(np.ones(n), pose, LASER.min_angle, LASER.ang_res)
    h.points = points.copy()                 # AddScan reads the point readings only (Mapper.cpp:1047-1105)
    return o, h


def _grids(preset, query_pose, base_points, batch_slot=0):
    om = make_oracle_matcher(preset)
    hm = make_hip_matcher(preset, max_batch=batch_slot + 1)
    oq, hq = _pair(np.zeros((1, 2)), query_pose)
    ob, hb = [], []
    for pts in base_points:
        o, h = _pair(pts, query_pose)
        ob.append(o); hb.append(h)
    om.add_scans(oq, ob)
    hm.AddScans(hq, hb, slot=batch_slot)
    want, got = om.grid().copy(), hm.GetCorrelationGrid(batch_slot)
    hm.close()
    return want, got


@pytest.mark.parametrize("preset", ["K", "L"])
def test_find_valid_points_corner_cases_through_the_kernel(kartohip_lib, preset):
    cases = list(_cases())
    occupied = 0
    for k, (points, view) in enumerate(cases):
        pose = np.array([view[0], view[1], 0.3])            # the viewpoint of AddScans is the query's sensor position
        want, got = _grids(preset, pose, [points])
        assert np.array_equal(want, got), f"case {k}: {int((want != got).sum())} cells differ"
        occupied += int((want == 100).sum())
    assert occupied > 1000
    # all of them as the base scans of ONE call (ragged lengths 1081 / 50 / 63, NaN-only scans in between)
    pose = np.array([1.0, -2.0, 0.3])
    want, got = _grids(preset, pose, [c[0] for c in cases])
    assert np.array_equal(want, got)


def test_find_valid_points_of_long_scans_takes_the_lane_kernel(kartohip_lib):
    """more than 2048 readings per scan: k_find_valid_lane (one lane walks the scan) instead of the LDS formulation"""
    rng = np.random.default_rng(12)
    pose = np.array([0.5, 0.25, -0.2])
    scans = []
    for n in (4000, 2049, 6001):
        ang = pose[2] + np.linspace(-2.3, 2.3, n)
        r = 5.0 + 1.5 * np.sin(np.linspace(0, 23, n)) + rng.normal(0, 0.01, n)
        r[rng.uniform(size=n) < 0.02] = np.nan
        r[rng.uniform(size=n) < 0.02] = np.inf
        r[n // 3:n // 3 + 150] = np.nan
        scans.append(np.stack([pose[0] + r * np.cos(ang), pose[1] + r * np.sin(ang)], axis=1))
    want, got = _grids("K", pose, scans)
    assert np.array_equal(want, got) and (want == 100).sum() > 2000
    # mixed with ordinary scans in the same call
    short = list(_cases())[0][0]
    want, got = _grids("K", pose, [short, scans[0], short[::-1].copy(), scans[1]])
    assert np.array_equal(want, got)


def _cells_to_points(cells, pose, res):
    """cell coordinates (relative to the query, which sits at a cell centre) -> point readings 0.3 of a cell off centre"""
    c = np.asarray(cells, dtype=np.float64)
    return np.stack([pose[0] + (c[:, 0] + 0.3) * res, pose[1] + (c[:, 1] - 0.2) * res], axis=1)


@pytest.mark.parametrize("preset,res", [("S", 0.01), ("L", 0.05)])
def test_occupied_cell_rule_corner_cases_through_the_kernels(kartohip_lib, preset, res):
    rng = np.random.default_rng(3)
    pose = np.array([2.0, -1.0, 0.7])
    block = [(x, y) for y in range(12) for x in range(12)]
    stairs = []
    for k in range(80):
        stairs += [(100 + k, 100 + k), (101 + k, 100 + k)]
    dense_row = [(k // 3 - 150, -40) for k in range(900)]              # three readings per cell, 300 cells in a row
    cases = {
        "block": [block], "stairs": [stairs], "reversed": [block[::-1] + stairs[::-1]],
        "row then block over it": [dense_row, [(x - 150, y - 46) for x, y in block]],
        "walls": [_walls(rng, 12) for _ in range(5)],                  # five scans revisiting each other's walls
        "walls, shifted copies": [[(x - 30, y - 30) for x, y in w] for w in [_walls(np.random.default_rng(8), 25)] * 3],
    }
    for name, scans in cases.items():
        pts = [_cells_to_points(cells, pose, res) for cells in scans]
        want, got = _grids(preset, pose, pts)
        assert np.array_equal(want, got), f"{name}: {int((want != got).sum())} cells differ"
        assert (want == 100).sum() > 20, name
    # the same through a batch slot other than 0
    pts = [_cells_to_points(c, pose, res) for c in cases["walls"]]
    want, got = _grids(preset, pose, pts, batch_slot=2)
    assert np.array_equal(want, got)


@pytest.mark.parametrize("case", range(3))
def test_range_built_corner_cases_against_the_reference_build(kartohip_lib, case):
    from oracle import ref
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref.init_laser(LASER)
    pose = np.array([2.0, 1.0, -0.4])
    ranges = _range_cases()[case]
    base_poses = [pose + np.array([0.05 * k, -0.03 * k, 0.01 * k]) for k in range(4)]
    rm = ref.RefMatcher(*PRESETS["S"]["create"], OFFLINE_PARAMS)
    rq = ref.RefScan(ranges, pose)
    rm.add_scans(rq, [ref.RefScan(np.roll(ranges, 17 * k), p) for k, p in enumerate(base_poses)])
    want = rm.grid().copy()
    hm = make_hip_matcher("S")
    hq = LocalizedRangeScan(ranges, pose, LASER.min_angle, LASER.ang_res)
    hm.AddScans(hq, [LocalizedRangeScan(np.roll(ranges, 17 * k), p, LASER.min_angle, LASER.ang_res) for k, p in enumerate(base_poses)])
    got = hm.GetCorrelationGrid()
    hm.close()
    assert np.array_equal(want, got) and (want == 100).sum() > 100
