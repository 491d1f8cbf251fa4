"""One MatchScan at a time (what the reference's API issues): wall time per call from Python through the C ABI, fused path of
the library against its general (batch) path, base scans resident and uploaded per call.  `--loop N preset` runs only the fused
loop (for rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from common import Scenario, make_hip_matcher  # noqa: E402


def timed(hm, q, b, n=200):
    for _ in range(5):
        hm.MatchScan(q, b, True, True)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        for _ in range(n // 5):
            hm.MatchScan(q, b, True, True)
        ts.append((time.perf_counter() - t) / (n // 5) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


if len(sys.argv) > 1 and sys.argv[1] == "--loop":
    n, preset = int(sys.argv[2]), sys.argv[3]
    general = len(sys.argv) > 4 and sys.argv[4] == "general"
    sc = Scenario(seed=11, n_base=20 if preset == "L" else 10, start=20)
    q, b = sc.hip_scans()
    for s in b:
        s.MakeResident(0)
    hm = make_hip_matcher(preset)
    if general:
        hm.set_debug(False, no_fused_match=True)
    for _ in range(n):
        hm.MatchScan(q, b, True, True)
    print(preset, hm.seq_stats())
    sys.exit(0)

for preset, nbase in (("S", 10), ("S", 20), ("L", 20), ("C2", 10), ("K", 10)):
    sc = Scenario(seed=11, n_base=nbase, start=20)
    for resident in (False, True):
        q, b = sc.hip_scans()
        if resident:
            for s in b:
                s.MakeResident(0)
        row = []
        for general in (False, True):
            hm = make_hip_matcher(preset)
            if general:
                hm.set_debug(False, no_fused_match=True)
            med, best = timed(hm, q, b)
            row.append((med, best, hm.seq_stats()))
            hm.close()
        print(f"{preset} n_base={nbase} resident={resident}: fused {row[0][0]:.3f} ms (best window {row[0][1]:.3f}), general {row[1][0]:.3f} ms "
              f"(best {row[1][1]:.3f}); fused stats {row[0][2]}", flush=True)
