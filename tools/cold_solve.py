import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_solver import HipSpaSolver
g = synth.make_pose_graph(10000, 30000, seed=12345)
n_odo = 9999
orders = (np.arange(len(g["edges"])), np.concatenate([np.arange(n_odo), np.arange(len(g["edges"]) - 1, n_odo - 1, -1)]))
sol = HipSpaSolver()
for rep in range(5):
    o = orders[rep % 2]
    sol.load(g["init"], g["edges"][o], g["z"][o], g["cov"][o])
    t = time.time(); s = sol.Compute(); dt = time.time() - t
    print("compute %.2f ms  solve_ms %.2f symbolic_ms %.2f total_ms %.2f analysis %d" % (dt * 1e3, s["solve_ms"], s["symbolic_ms"], s["total_ms"], s["analysis"]), file=sys.stderr)
