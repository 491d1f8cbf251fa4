import sys, os, time, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import LASER, OFFLINE_PARAMS, PRESETS
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcher, _scan_array
world = synth.make_world(12345)
truth, _ = synth.trajectory(2000)
rng = np.random.default_rng(99)
distinct, batch = 32, 64
queries, chains = [], []
for k in range(distinct):
    q = 150 + 53 * k
    d = np.hypot(truth[:, 0] - truth[q, 0], truth[:, 1] - truth[q, 1]); d[max(0, q - 80): q + 80] = 1e9
    j = int(np.argmin(d)); length = 10 + (7 * k) % 31
    lo = max(0, min(len(truth) - length, j - length // 2))
    chains.append([LocalizedRangeScan(synth.make_scan(world, truth[i], rng), truth[i], LASER.min_angle, LASER.ang_res) for i in range(lo, lo + length)])
    queries.append(LocalizedRangeScan(synth.make_scan(world, truth[q], rng), truth[q] + np.array([0.15 * math.sin(k), -0.1 * math.cos(k), 0.03 * ((k % 5) - 2)]), LASER.min_angle, LASER.ang_res))
mp = MapperParams(**OFFLINE_PARAMS)
for name, pen, ref in (("L", False, False), ("S", False, True)):
    m = ScanMatcher.Create(mp, *PRESETS[name]["create"], device=0, max_batch=batch)
    ids = [i % distinct for i in range(batch)]
    qs = [queries[i] for i in ids]; cs = [chains[i] for i in ids]
    m.MatchScanBatch(qs, cs, pen, ref)
    m.profile(True)
    t = time.perf_counter()
    for _ in range(4): m.MatchScanBatch(qs, cs, pen, ref)
    dt = (time.perf_counter() - t) / 4
    pr = m.profile(False)
    t = time.perf_counter()
    flat = [b for lst in cs for b in lst]; _scan_array(qs); _scan_array(flat)
    tm = time.perf_counter() - t
    print(name, "batch of", batch, "ms %.2f" % (dt * 1e3), "marshalling ms %.2f" % (tm * 1e3), {k: (v / 4 if isinstance(v, float) else v) for k, v in pr.items()})
    m.close()
