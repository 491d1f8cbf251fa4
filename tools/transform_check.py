import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_toolbox_amd import synth
from slam_toolbox_amd.mapper import Mapper
n_scans = 900
world = synth.make_world(12345)
truth, odom = synth.trajectory_laps(n_scans)
rng = np.random.default_rng(4)
ranges = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], rng) for i in range(n_scans)]))
os.environ["KH_MAPPER_DUMP_MATCH"] = "514:/tmp/m514.bin"
m = Mapper(synth.Laser(), loop_search_maximum_distance=3.0)
q_of = []
final = {}
for i in range(n_scans):
    ok, pose, _ = m.Process(ranges[i], odom[i], 0.1 * i)
    if ok:
        q_of.append(i)
        final[len(q_of) - 1] = m.poses()[len(q_of) - 1].copy()
    if len(q_of) == 515:
        break
raw = np.fromfile("/tmp/m514.bin"); qpose = raw[2:5]
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libkarto_ref.so"))
lib.ref_transform_pose.argtypes = [C.c_void_p] * 4
out = np.zeros(3)
p1 = np.ascontiguousarray(odom[q_of[513]]); p2 = np.ascontiguousarray(final[513]); src = np.ascontiguousarray(odom[q_of[514]])
lib.ref_transform_pose(p1.ctypes.data, p2.ctypes.data, src.ctypes.data, out.ctypes.data)
print("odom 513", p1.tolist()); print("corrected 513", p2.tolist()); print("odom 514", src.tolist())
print("kh_mapper predicted pose :", qpose.tolist())
print("reference Transform      :", out.tolist())
print("identical:", np.array_equal(qpose.view(np.uint64), out.view(np.uint64)))
