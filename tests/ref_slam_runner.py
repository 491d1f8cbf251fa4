"""Helper process of tests/test_dropin_mapper_gpu.py: runs the synthetic scan queue through the reference's own
karto::Mapper in ONE of the oracle/_ref builds (each build carries its own copy of the karto singletons, so each
run gets a process of its own).  usage: ref_slam_runner.py <lib.so> <n_scans> <loop_search_distance> <out_prefix> [sweep|laps]
[removal schedule or ""] [laser mount offset "x,y,heading"]
sweep = one boustrophedon pass through the aisles (4 m apart: needs a 5 m loop search distance to close anything);
laps = the same two aisles driven lap after lap (closes loops with the shipped 3.0 m, offline.yaml:40)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_toolbox_amd import synth  # noqa: E402


def main():
    lib_path, n_scans, loop_dist, prefix = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
    kind = sys.argv[5] if len(sys.argv) > 5 else "sweep"
    # optional removal schedule "at:id,at:id,...": after queue scan `at` the node of scan `id` is removed (lifelong graph edits)
    schedule = [tuple(int(v) for v in item.split(":")) for item in sys.argv[6].split(",")] if len(sys.argv) > 6 and sys.argv[6] else []
    lib = C.CDLL(lib_path)
    lib.ref_init_laser.restype = C.c_int
    lib.ref_init_laser.argtypes = [C.c_double] * 6
    lib.ref_slam_run.restype = C.c_int
    lib.ref_slam_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_char_p, C.c_void_p, C.c_int]
    laser = synth.Laser()
    n_beams = lib.ref_init_laser(laser.min_angle, laser.max_angle, laser.ang_res, laser.min_range, laser.max_range,
                                 laser.range_threshold)
    if len(sys.argv) > 7 and sys.argv[7]:
        lib.ref_set_laser_offset.argtypes = [C.c_double] * 3
        assert lib.ref_set_laser_offset(*[float(v) for v in sys.argv[7].split(",")]) == 0
    lib.ref_set_threads(min(32, os.cpu_count() or 1))
    world = synth.make_world(12345)
    truth, odom = synth.trajectory_laps(n_scans) if kind == "laps" else synth.trajectory(n_scans)
    rng = np.random.default_rng(4)
    ranges = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], rng) for i in range(n_scans)]))
    assert ranges.shape[1] == n_beams
    odom = np.ascontiguousarray(odom)
    out = np.zeros((n_scans, 4))
    t0 = time.perf_counter()
    if schedule:
        lib.ref_slam_run_schedule.restype = C.c_int
        lib.ref_slam_run_schedule.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_char_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int]
        at = np.ascontiguousarray([a for a, _ in schedule], dtype=np.int32)
        ids = np.ascontiguousarray([i for _, i in schedule], dtype=np.int32)
        accepted = lib.ref_slam_run_schedule(n_scans, n_beams, ranges.ctypes.data, odom.ctypes.data, loop_dist,
                                             (prefix + ".log").encode(), out.ctypes.data, n_scans, at.ctypes.data, ids.ctypes.data,
                                             len(schedule))
    else:
        accepted = lib.ref_slam_run(n_scans, n_beams, ranges.ctypes.data, odom.ctypes.data, loop_dist,
                                    (prefix + ".log").encode(), out.ctypes.data, n_scans)
    seconds = time.perf_counter() - t0
    calls = -1
    if hasattr(lib, "ref_gpu_matcher_calls"):
        lib.ref_gpu_matcher_calls.restype = C.c_long
        calls = lib.ref_gpu_matcher_calls()
        lib.ref_gpu_matcher_release()
    np.savez(prefix + ".npz", accepted=accepted, poses=out[:max(accepted, 0)], seconds=seconds, gpu_matcher_calls=calls)


if __name__ == "__main__":
    main()
