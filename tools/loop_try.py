"""Loop-closure candidate leg of bench.py at other batch sizes (diagnostics).  usage: python tools/loop_try.py 64 256"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
for b in [int(v) for v in sys.argv[1:]] or [64, 128, 256]:
    r = bench.loop_leg(0, batch=b)
    print(b, r["loop_pairs_per_s"], r["loop_batch_ms"])
