"""kh_loop_closure_batch on the bench's 256-pair batch for several piece counts."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import LASER, OFFLINE_PARAMS, PRESETS
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, LoopClosureBatch, MapperParams, ScanMatcher
lb = synth.loop_batch(256)
cache = {}
def scan_at(i):
    if i not in cache:
        cache[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res)
        cache[i].MakeResident(0)
    return cache[i]
queries = [LocalizedRangeScan(lb["ranges"][q], pose, LASER.min_angle, LASER.ang_res) for q, pose, _ in lb["pairs"]]
chains = [[scan_at(i) for i in chain] for _, _, chain in lb["pairs"]]
mp = MapperParams(**OFFLINE_PARAMS)
mL = ScanMatcher.Create(mp, *PRESETS["L"]["create"], device=0, max_batch=256)
mS = ScanMatcher.Create(mp, *PRESETS["S"]["create"], device=0, max_batch=256)
pack = ScanMatcher.pack_batch(queries, chains)
for pieces in [int(x) for x in os.environ.get("PIECES", "1,1,2,3,4,6,8").split(",")]:
    LoopClosureBatch(mL, mS, None, None, LASER.min_angle, LASER.ang_res, 0.35, 9.0, pieces=pieces, packed=pack)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        LoopClosureBatch(mL, mS, None, None, LASER.min_angle, LASER.ang_res, 0.35, 9.0, pieces=pieces, packed=pack)
        ts.append(time.perf_counter() - t)
    print(pieces, "pieces:", round(float(np.median(ts)) * 1e3, 3), "ms", flush=True)
