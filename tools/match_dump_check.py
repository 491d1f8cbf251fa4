"""Replays a sequential match dumped by kh_mapper (KH_MAPPER_DUMP_MATCH=<id>:<path>) through the HIP matcher, the C oracle and
the reference build, and compares the three results bit by bit: python tools/match_dump_check.py <dump>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import LASER, OFFLINE_PARAMS, PRESETS, bits, make_hip_matcher, make_oracle_matcher
raw = np.fromfile(sys.argv[1])
nb, n = int(raw[0]), int(raw[1]); p = 2
qpose = raw[p:p + 3]; p += 3
qr = raw[p:p + n]; p += n
base = []
for _ in range(nb):
    pose = raw[p:p + 3]; p += 3
    base.append((pose.copy(), raw[p:p + n].copy())); p += n
mean, cov, resp = raw[p:p + 3], raw[p + 3:p + 12], raw[p + 12]
print("dumped result:", mean.tolist(), resp)
from oracle import karto
from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
om = make_oracle_matcher("S"); hm = make_hip_matcher("S")
oq = karto.Scan(qr, qpose, LASER); ob = [karto.Scan(r, ps, LASER) for ps, r in base]
hq = LocalizedRangeScan(qr, qpose, LASER.min_angle, LASER.ang_res); hb = [LocalizedRangeScan(r, ps, LASER.min_angle, LASER.ang_res) for ps, r in base]
ro, mo, co = om.match_scan(oq, ob, True, True)
rh, mh, ch = hm.MatchScan(hq, hb, True, True)
print("oracle :", np.asarray(mo).tolist(), ro)
print("hip    :", np.asarray(mh).tolist(), rh)
print("hip == dump:", np.array_equal(bits(mh), bits(mean)), " hip == oracle:", np.array_equal(bits(mh), bits(mo)), np.array_equal(bits(ch), bits(co)), rh == ro)
from oracle import ref
if ref.available():
    ref.init_laser(LASER)
    rm = ref.RefMatcher(*PRESETS["S"]["create"], OFFLINE_PARAMS)
    rr, mr, cr = rm.match_scan(ref.RefScan(qr, qpose), [ref.RefScan(r, ps) for ps, r in base], True, True)
    print("ref    :", np.asarray(mr).tolist(), rr, " ref == oracle:", np.array_equal(bits(mr), bits(mo)), " ref == hip:", np.array_equal(bits(mr), bits(mh)))
hm.close()
