"""The loop-closure leg of bench.py on its own (BASELINE config[2]): python tools/loop_leg.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
out = bench.loop_leg(cpu=False)
print(json.dumps({k: v for k, v in out.items() if k in ("loop_batch_ms", "loop_gpu_ms", "loop_rooflines", "loop_batch_ms_host_scans")}))
