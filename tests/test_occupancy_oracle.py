"""CPU: the occupancy-grid oracle (oracle/occupancy_oracle.c) against the reference's own
OccupancyGrid::CreateFromScans output (tests/golden/occupancy.npz): cell states and both counter grids."""
import os

import numpy as np

from common import LASER
from oracle import karto

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "occupancy.npz"))


def golden_dense():
    w, h, ws = (int(v) for v in G["dims"])
    cells = np.zeros(ws * h, dtype=np.uint8)
    cells[G["cells_idx"]] = G["cells_val"]
    passes = np.zeros(ws * h, dtype=np.uint32)
    hits = np.zeros(ws * h, dtype=np.uint32)
    passes[G["count_idx"]] = G["pass_val"]
    hits[G["count_idx"]] = G["hit_val"]
    return w, h, ws, cells.reshape(h, ws), passes.reshape(h, ws), hits.reshape(h, ws)


def test_oracle_reproduces_the_reference_grid(oracle_lib):
    w, h, ws, cells, passes, hits = golden_dense()
    scans = [karto.Scan(G["ranges"][k], G["poses"][k], LASER) for k in range(G["ranges"].shape[0])]
    c, p, hh = karto.occupancy_from_scans(w, h, G["offset"], float(G["resolution"]), scans, LASER)
    assert np.array_equal(p, passes)
    assert np.array_equal(hh, hits)
    assert np.array_equal(c, cells)
    assert (c == 100).sum() > 100 and (c == 255).sum() > 10000


def test_compute_dimensions_matches_the_reference(kartohip_lib):
    """kh_occupancy_compute_dimensions is host arithmetic (no device needed): OccupancyGrid::ComputeDimensions."""
    from slam_toolbox_amd.occupancy_grid import compute_dimensions
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    scans = [LocalizedRangeScan(G["ranges"][k], G["poses"][k], LASER.min_angle, LASER.ang_res) for k in range(G["ranges"].shape[0])]
    w, h, off = compute_dimensions(scans, LASER.min_range, LASER.range_threshold, float(G["resolution"]))
    assert (w, h) == (int(G["dims"][0]), int(G["dims"][1]))
    assert np.array_equal(off.view(np.uint64), np.asarray(G["offset"], dtype=np.float64).view(np.uint64))
