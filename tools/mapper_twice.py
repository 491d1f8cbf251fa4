"""kh_mapper on the 2000-scan lap queue, solver-call log to argv[1] (determinism check: run twice, diff the logs)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_toolbox_amd import synth
from slam_toolbox_amd.mapper import Mapper
n_scans = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
world = synth.make_world(12345)
truth, odom = synth.trajectory_laps(n_scans)
rng = np.random.default_rng(4)
ranges = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], rng) for i in range(n_scans)]))
m = Mapper(synth.Laser(), loop_search_maximum_distance=3.0, log_path=sys.argv[1])
for i in range(n_scans):
    m.Process(ranges[i], odom[i], 0.1 * i)
m.set_log(None); m.close()
