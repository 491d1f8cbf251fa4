"""Preset L coarse scoring of the bench's loop batch under the debug switches: time per batch, wave-level loads, clocks per load."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import LASER, OFFLINE_PARAMS, PRESETS
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcher
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
for label, kw in (("copies", {}),):
    m = ScanMatcher.Create(mp, *PRESETS["L"]["create"], device=0, max_batch=256)
    m.set_debug(False, **kw)
    pack = ScanMatcher.pack_batch(queries, chains)
    m.MatchScanBatch(queries, chains, False, False, packed=pack)
    m.profile(True)
    m.score_loads()
    t = time.perf_counter()
    for _ in range(3):
        m.MatchScanBatch(queries, chains, False, False, packed=pack)
    dt = (time.perf_counter() - t) / 3
    pr = m.profile(False)
    loads = m.score_loads() / 3
    ms = pr["score_ms"] / 3
    print(label, json.dumps({"batch_ms": dt * 1e3, "score_ms": ms, "wave_loads": loads, "clocks_per_load": ms * 1e-3 * 2.4e9 * 256 / max(1, loads),
                             "raster_ms": pr["raster_ms"] / 3}), flush=True)
    m.close()
