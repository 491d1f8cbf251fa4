"""GPU parity of the occupancy grid (kh_occupancy_*, through the C ABI): cell states and both counter grids
bit-exact against the reference's own OccupancyGrid::CreateFromScans (tests/golden/occupancy.npz) and, on 300
scans at 2.5 cm, against the CPU oracle."""
import numpy as np
import pytest

from common import LASER
from slam_toolbox_amd import synth
from test_occupancy_oracle import G, golden_dense

pytestmark = pytest.mark.gpu


def _hip_scans(ranges, poses):
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    return [LocalizedRangeScan(ranges[k], poses[k], LASER.min_angle, LASER.ang_res) for k in range(len(ranges))]


def test_grid_matches_the_reference(kartohip_lib):
    from slam_toolbox_amd.occupancy_grid import OccupancyGrid
    w, h, ws, cells, passes, hits = golden_dense()
    g = OccupancyGrid.CreateFromScans(_hip_scans(G["ranges"], G["poses"]), float(G["resolution"]), LASER)
    assert (g.GetWidth(), g.GetHeight(), g.width_step) == (w, h, ws)
    p, hh = g.counters()
    assert np.array_equal(p, passes) and np.array_equal(hh, hits)
    assert np.array_equal(g.cells(), cells)
    # adding the same scans in two batches gives the same counters (increments commute)
    g.Clear()
    scans = _hip_scans(G["ranges"], G["poses"])
    g.AddScans(scans[:10], LASER)
    g.AddScans(scans[10:], LASER)
    g.Update()
    assert np.array_equal(g.cells(), cells)
    g.close()
    assert OccupancyGrid.CreateFromScans([], 0.05, LASER) is None


def test_large_map_against_the_oracle(kartohip_lib, oracle_lib):
    from oracle import karto
    from slam_toolbox_amd.occupancy_grid import OccupancyGrid
    world = synth.make_world(12345)
    truth, _ = synth.trajectory(900)
    rng = np.random.default_rng(8)
    idx = list(range(0, 900, 3))
    ranges = [synth.make_scan(world, truth[i], rng) for i in idx]
    poses = truth[idx]
    g = OccupancyGrid.CreateFromScans(_hip_scans(ranges, poses), 0.025, LASER, min_pass_through=3, occupancy_threshold=0.2)
    oscans = [karto.Scan(ranges[k], poses[k], LASER) for k in range(len(idx))]
    c, p, hh = karto.occupancy_from_scans(g.width, g.height, g.offset, 0.025, oscans, LASER, 3, 0.2)
    gp, gh = g.counters()
    assert np.array_equal(gp, p) and np.array_equal(gh, hh)
    assert np.array_equal(g.cells(), c)
    assert (c == 100).sum() > 1000
    g.close()
