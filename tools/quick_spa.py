import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_toolbox_amd import synth, capi
from slam_toolbox_amd.scan_solver import HipSpaSolver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
e = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
g = synth.make_pose_graph(n, e, seed=12345)
sol = HipSpaSolver()
if "--phases" in sys.argv:
    sol.set_debug(phase_timing=True)
for rep in range(3):
    t = time.time(); sol.load(g["init"], g["edges"], g["z"], g["cov"]); tl = time.time() - t
    t = time.time(); summ = sol.Compute(); tc = time.time() - t
    print("load %.1f ms compute %.1f ms" % (tl * 1e3, tc * 1e3), summ, sol.last_warning, capi.lib().kh_last_error())
if "--check" in sys.argv:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import spa
    t = time.time(); x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"]); print("oracle s", time.time() - t, info["iterations"], info["final_cost"])
    d = sol.poses() - x; d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    print("max diff vs oracle", np.abs(d).max())
