"""Generates the golden fixtures in this directory from the REFERENCE ITSELF: the reference's own
karto_sdk sources compiled in place (oracle/_ref/libkarto_ref.so, see oracle/Makefile).  Run in the dev
container (where /root/reference exists):

    make -C oracle ref && python tests/golden/make_golden.py

Each fixture holds the inputs (ranges, poses, parameters) and the reference's outputs for one scenario:
smear kernel, FindValidPoints output, rasterised grid (non-zero cells), lookup table, search-space
probabilities, raw GetResponse values, and MatchScan / CorrelateScan results.  The fixtures are what
pins oracle/karto_oracle.c and the HIP path on machines without the reference (the GPU box)."""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import LASER, PRESETS, Scenario  # noqa: E402
from oracle import ref  # noqa: E402


def sparse(grid):
    idx = np.flatnonzero(grid).astype(np.int32)
    return idx, grid[idx]


def make_fixture(name, preset, seed, n_base, start, perturb, correlate=None):
    sc = Scenario(seed=seed, n_base=n_base, start=start, perturb=perturb)
    q, base = sc.ref_scans()
    p = PRESETS[preset]
    m = ref.RefMatcher(*p["create"], p["params"])
    out = dict(preset=preset, create=np.asarray(p["create"]), seed=seed, n_base=n_base, start=start,
               perturb=np.asarray(perturb), query_pose=sc.query_pose, base_poses=np.asarray(sc.base_poses),
               ranges=np.asarray(sc.ranges), kernel=m.kernel())
    gi = m.grid_info()
    out["grid_geom"] = np.asarray([gi[k] for k in ("width", "height", "width_step", "roi_x", "roi_y", "roi_w", "roi_h",
                                                   "kernel_size", "data_size")], dtype=np.int64)
    # points as the reference's Update() makes them + FindValidPoints of base scan 0
    out["query_points"] = q.points()
    out["valid_points_0"] = m.find_valid_points(base[0], sc.query_pose[:2])
    # MatchScan for the three flag combinations the mapper uses
    res = []
    for pen, refine in [(True, True), (False, True), (False, False)]:
        r, mean, cov = m.match_scan(q, base, pen, refine)
        res.append(np.concatenate([[r], mean, cov.reshape(9)]))
    out["match_results"] = np.asarray(res)
    gi = m.grid_info()
    out["grid_offset"] = np.asarray([gi["offset_x"], gi["offset_y"], gi["scale"]])
    idx, val = sparse(m.grid())
    out["grid_idx"], out["grid_val"] = idx, val
    # an explicit CorrelateScan on that grid: lookup table, probs, raw responses, result
    if correlate is None:
        resolution = 1.0 / gi["scale"]
        side = p["create"][0]
        off = 0.5 * round(side / resolution) * resolution
        correlate = dict(off=(off, off), res=(2 * resolution, 2 * resolution),
                         ang_off=p["params"]["coarse_search_angle_offset"], ang_res=p["params"]["coarse_angle_resolution"],
                         penalize=True, fine=False)
    r, mean, cov = m.correlate_scan(q, sc.query_pose, correlate["off"], correlate["res"], correlate["ang_off"],
                                    correlate["ang_res"], correlate["penalize"], correlate["fine"])
    out["correlate_args"] = np.asarray([*correlate["off"], *correlate["res"], correlate["ang_off"], correlate["ang_res"],
                                        float(correlate["penalize"]), float(correlate["fine"])])
    out["correlate_result"] = np.concatenate([[r], mean, cov.reshape(9)])
    na = int(ref.lib().ref_lookup_angles(m.h))
    out["lookup"] = m.lookup_table(na, LASER.n_beams)
    if not correlate["fine"]:
        out["probs"] = m.probs(int(round(p["create"][0] / p["create"][1])) + 1)
    # raw GetResponse (Mapper.cpp:1172-1208) at the search centre and at two offset cells for every angle
    cells = [m.world_to_grid_index(sc.query_pose[0] + dx, sc.query_pose[1] + dy) for dx, dy in
             [(0.0, 0.0), (correlate["res"][0], 0.0), (-correlate["res"][0], correlate["res"][1])]]
    out["response_cells"] = np.asarray(cells, dtype=np.int32)
    out["raw_responses"] = np.asarray([[m.get_response(a, c) for a in range(na)] for c in cells])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, os.path.getsize(path) // 1024, "KiB", "match responses", out["match_results"][:, 0], "corr", r)


def main():
    ref.init_laser(LASER)
    make_fixture("match_K", "K", seed=11, n_base=10, start=20, perturb=(0.05, -0.03, 0.02))
    make_fixture("match_S", "S", seed=12, n_base=10, start=120, perturb=(-0.06, 0.04, -0.03))
    make_fixture("match_L", "L", seed=13, n_base=20, start=200, perturb=(0.6, -0.9, 0.1))
    make_fixture("corr_C2", "C2", seed=7, n_base=10, start=0, perturb=(0.05, -0.03, 0.02),
                 correlate=dict(off=(0.15, 0.15), res=(0.005, 0.005), ang_off=math.radians(20.0),
                                ang_res=math.radians(0.5), penalize=True, fine=False))
    # solver-side known answers from the reference's own classes (LinkInfo::Update, Matrix3::Inverse)
    rng = np.random.default_rng(99)
    p1 = rng.uniform(-5, 5, size=(16, 3))
    p2 = rng.uniform(-5, 5, size=(16, 3))
    covs, diffs, couts, invs = [], [], [], []
    for i in range(16):
        a = rng.normal(size=(3, 3))
        c = a @ a.T * 1e-3 + np.eye(3) * 1e-4
        d, co = ref.link_info(p1[i], p2[i], c)
        covs.append(c); diffs.append(d); couts.append(co); invs.append(ref.matrix3_inverse(co))
    np.savez_compressed(os.path.join(HERE, "link_info.npz"), pose1=p1, pose2=p2, cov=np.asarray(covs),
                        diff=np.asarray(diffs), cov_out=np.asarray(couts), inverse=np.asarray(invs))


if __name__ == "__main__":
    main()
