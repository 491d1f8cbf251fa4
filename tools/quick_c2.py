import sys, time, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from common import *
from slam_toolbox_amd import capi
print("devices", capi.lib().kh_device_count())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
preset = sys.argv[2] if len(sys.argv) > 2 else "C2"
sc = Scenario(seed=7, n_base=10, start=0)
hq, hb = sc.hip_scans()
hm = make_hip_matcher(preset, max_batch=B)
for s in range(B):
    hm.AddScans(hq, hb, slot=s)
if preset == "C2":
    args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
else:
    res = 1.0 / hm.grid_info()["scale"]; p = PRESETS[preset]["params"]; side = PRESETS[preset]["create"][0]
    off = 0.5 * round(side / res) * res
    args = ((off, off), (2 * res, 2 * res), p["coarse_search_angle_offset"], p["coarse_angle_resolution"])
from slam_toolbox_amd.scan_matcher import _scan_array
arr = (_scan_array([hq] * B), B)
centers = np.tile(sc.query_pose, (B, 1))
for it in range(3):
    hm.CorrelateScanBatch(None, centers, *args, True, False, scan_array=arr)
hm.profile(True)
t = time.time(); N = 20
for it in range(N):
    r = hm.CorrelateScanBatch(None, centers, *args, True, False, scan_array=arr)
dt = (time.time() - t) / N
prof = hm.profile(False)
print("batch", B, "wall ms/step", dt * 1e3, "matches/s", B / dt, "resp", r[0][0], prof, "score ms/launch", prof["score_ms"] / max(1, prof["score_launches"]))
t = time.time()
for it in range(5):
    for s in range(B):
        hm.AddScans(hq, hb, slot=s)
print("addscans ms each", (time.time() - t) / 5 / B * 1e3)
t = time.time()
r = hm.MatchScanBatch([hq] * B, [hb] * B)
print("matchscan batch ms", (time.time() - t) * 1e3, r[0][:2])
