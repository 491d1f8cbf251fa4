"""One sequential-preset MatchScan at a time (the mapper's per-scan call): wall time per call, KH_MATCH_TIMING split, and -- under
rocprofv3 --kernel-trace -- the kernel timeline of one call: python tools/seq_latency.py [n_base] [resident]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import Scenario, make_hip_matcher  # noqa: E402
nbase = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sc = Scenario(seed=11, n_base=nbase, start=20)
q, b = sc.hip_scans()
if len(sys.argv) > 2:
    for s in b:
        s.MakeResident()
hm = make_hip_matcher("S")
for _ in range(3):
    hm.MatchScan(q, b, True, True)
t = time.perf_counter()
n = 50
for _ in range(n):
    r = hm.MatchScan(q, b, True, True)
print("S MatchScan, %d base scans%s: %.3f ms per call, response %.4f" % (nbase, " (resident)" if len(sys.argv) > 2 else "", (time.perf_counter() - t) / n * 1e3, r[0]))
hm.close()
