import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import PRESETS, Scenario, make_hip_matcher
for preset, nbase in (("S", 10), ("L", 20), ("C2", 10), ("K", 10)):
    sc = Scenario(seed=11, n_base=nbase, start=20)
    q, b = sc.hip_scans()
    hm = make_hip_matcher(preset)
    for flags in ((True, True), (False, False)):
        hm.MatchScan(q, b, *flags)
        t = time.perf_counter()
        for _ in range(20):
            r = hm.MatchScan(q, b, *flags)
        dt = (time.perf_counter() - t) / 20
        print(preset, "MatchScan penalize/refine", flags, "%.3f ms" % (dt * 1e3), "response %.4f" % r[0])
    hm.close()
