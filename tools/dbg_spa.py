import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_solver import HipSpaSolver
from oracle import spa
from test_spa_gpu import _clique_graph, _diff
which = sys.argv[1]
if which == "clique":
    g = _clique_graph(int(sys.argv[2]), int(sys.argv[3]), seed=3)
else:
    g = synth.make_pose_graph(int(sys.argv[2]), int(sys.argv[3]), seed=1)
sol = HipSpaSolver()
sol.load(g["init"], g["edges"], g["z"], g["cov"])
summ = sol.Compute()
print(summ, sol.last_warning)
x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
print("oracle iters", info["iterations"], "diff", _diff(sol.poses(), x))
