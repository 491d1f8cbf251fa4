"""The solver half against Ceres itself -- the pin DESIGN.md section 6 says is missing.  Ceres is not installed in the build
image, so this test SKIPS there; on a box with libceres-dev `make -C oracle ceres` builds oracle/_ref/ceres_driver (the
reference's own cost functor and options around ceres::Solve) and the test lays kh_spa_compute beside it iteration by
iteration: accepted / rejected steps, cost after every accepted step, trust-region radius, and the final poses within the
BASELINE tolerance (1e-4 m / 1e-4 rad)."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ceres_driver")


def _driver():
    if not os.path.exists(DRIVER):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ceres"], capture_output=True)
    return DRIVER if os.path.exists(DRIVER) else None


def _compare(path, kartohip_lib):
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    out = json.loads(subprocess.run([DRIVER, path], capture_output=True, text=True, check=True).stdout)
    sol = HipSpaSolver()
    sol.load_graph(path)
    summ = sol.Compute()
    log = sol.iteration_log()
    got = {i: p for i, p in sol.GetCorrections()}
    sol.close()
    assert out["usable"] == 1 and summ["usable"] == 1
    # Ceres row 0 is the initial evaluation; rows 1.. are the trust-region iterations
    rows = [r for r in out["iterations"] if r["iteration"] > 0]
    assert len(rows) == len(log), f"Ceres took {len(rows)} iterations, the library {len(log)}"
    for r, mine in zip(rows, log):
        accepted = mine[7] == 1.0
        assert bool(r["successful"]) == accepted, f"iteration {r['iteration']}: accepted differs"
        assert abs(r["radius"] - mine[5]) <= 1e-9 * abs(r["radius"]), f"iteration {r['iteration']}: radius {r['radius']} vs {mine[5]}"
        if accepted:
            assert abs(r["cost"] - mine[2]) <= 1e-9 * max(1.0, abs(r["cost"])), f"iteration {r['iteration']}: cost"
    for node_id, x, y, t in out["poses"]:
        d = got[int(node_id)] - np.array([x, y, t])
        d[2] = (d[2] + np.pi) % (2 * np.pi) - np.pi
        assert abs(d[0]) < 1e-4 and abs(d[1]) < 1e-4 and abs(d[2]) < 1e-4, f"node {node_id}: {d}"


@pytest.mark.gpu
def test_small_graph_matches_ceres(kartohip_lib):
    if _driver() is None:
        pytest.skip("Ceres is not installed here (no ceres/ceres.h): `make -C oracle ceres` builds the driver where it is; "
                    "until then the solver's parity is pinned only to oracle/spa.py (DESIGN.md section 6)")
    _compare(os.path.join(ROOT, "tests", "golden", "posegraph_small.g2o"), kartohip_lib)


@pytest.mark.gpu
def test_baseline_graph_matches_ceres(kartohip_lib, tmp_path):
    if _driver() is None:
        pytest.skip("Ceres is not installed here (no ceres/ceres.h): see test_small_graph_matches_ceres")
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(10000, 30000, seed=12345)
    sol = HipSpaSolver()
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    path = str(tmp_path / "config4.g2o")
    sol.save_graph(path, binary=False)
    sol.close()
    _compare(path, kartohip_lib)
