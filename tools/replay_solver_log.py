"""Replays the solver calls of a mapper log (N / C / E / D / K / X lines, tools/difflog-style logs written with Mapper(log_path=...))
into a fresh HipSpaSolver and compares every Compute()'s corrections with the log's P lines:
python tools/replay_solver_log.py <log> [check]   (check: kh_spa_set_debug bit 0 -- residual + clean-front checks)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from slam_toolbox_amd.scan_solver import HipSpaSolver
sol = HipSpaSolver()
if len(sys.argv) > 2:
    sol.set_debug(check_linear_solves=True)
lines = open(sys.argv[1]).read().split("\n")
i, ncomp = 0, 0
while i < len(lines):
    t = lines[i].split()
    i += 1
    if not t:
        continue
    if t[0] == "N":
        sol.AddNode(int(t[1]), [float(v) for v in t[2:5]])
    elif t[0] == "C":
        v = [float(x) for x in t[3:]]
        sol.AddConstraint(int(t[1]), int(t[2]), v[:3], np.array(v[3:12]).reshape(3, 3))
    elif t[0] == "E":
        sol.RemoveConstraint(int(t[1]), int(t[2]))
    elif t[0] == "D":
        sol.RemoveNode(int(t[1]))
    elif t[0] == "K":
        sol.Clear()
    elif t[0] == "X":
        summ = sol.Compute()
        corr = sol.GetCorrections()
        want = {}
        while i < len(lines) and lines[i].startswith("P "):
            p = lines[i].split(); want[int(p[1])] = [float(x) for x in p[2:5]]; i += 1
        got = {int(a): np.asarray(b) for a, b in corr}
        d = max(np.abs(np.array(want[k]) - got[k]).max() for k in want) if want else 0.0
        print(f"compute {ncomp} nodes {len(want)} iterations {summ['iterations']} analysis {summ['analysis']} max |dp| vs log {d:.3e} resid {summ['worst_linear_residual']:.2e}")
        if os.environ.get("REPLAY_ITER_LOG") and ncomp == int(os.environ["REPLAY_ITER_LOG"]):
            for row in sol.iteration_log():
                print("   ", " ".join(f"{v:.17g}" for v in row))
        ncomp += 1
        if ncomp >= 12:
            break
