import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import PRESETS, Scenario, bits, make_hip_matcher, make_oracle_matcher
preset = sys.argv[1]; fine = sys.argv[2] == "fine"
sc = Scenario(seed=5, n_base=8, start=60, perturb=(-0.04, 0.06, -0.03))
oq, ob = sc.oracle_scans(); hq, hb = sc.hip_scans()
om = make_oracle_matcher(preset, threads=8); hm = make_hip_matcher(preset)
hm.set_debug(True)
om.add_scans(oq, ob); hm.AddScans(hq, hb)
res = 1.0 / om.grid_info()["scale"]
p = PRESETS[preset]["params"]
if preset == "C2":
    args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
elif fine:
    args = ((res, res), (res, res), 0.5 * p["coarse_angle_resolution"], p["fine_search_angle_offset"])
else:
    side = PRESETS[preset]["create"][0]; off = 0.5 * round(side / res) * res
    args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
r_o, mean_o, cov_o = om.correlate_scan(oq, sc.query_pose, *args, False, fine)
r_h, mean_h, cov_h = hm.CorrelateScan(hq, sc.query_pose, *args, False, None, fine)
vol = om.volume(); sums, resp = hm.volume()
osum = np.rint(vol[..., 0] * (1081 * 100)).astype(np.int64)
bad = np.argwhere(osum != sums)
print(preset, "fine" if fine else "coarse", "shape", sums.shape, "mismatches", len(bad), "of", sums.size)
if len(bad):
    print("by angle:", np.bincount(bad[:, 2], minlength=sums.shape[2]))
    print("by y:", np.bincount(bad[:, 0], minlength=sums.shape[0]))
    print("by x:", np.bincount(bad[:, 1], minlength=sums.shape[1]))
    for b in bad[:8]:
        print(tuple(b), "oracle", osum[tuple(b)], "hip", sums[tuple(b)])
