"""Runs ONE leg of bench.py (for rocprofv3 --kernel-trace --stats runs per leg): python tools/prof_legs.py loop|solver|enum"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

leg = sys.argv[1]
if leg == "loop":
    out = bench.loop_leg(0, cpu=False)
elif leg == "solver":
    out = bench.solver_leg(0, cpu=False)
else:
    out = bench.enumeration_leg(0)
print(json.dumps(out))
