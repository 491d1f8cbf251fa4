"""BASELINE config 5 from the command line: python tools/replay.py --scans 50000 [--no-lifelong] [--mode async --period 0.025]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_toolbox_amd import replay  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=5000)
ap.add_argument("--no-lifelong", action="store_true")
ap.add_argument("--mode", default="sync", choices=["sync", "async"])
ap.add_argument("--period", type=float, default=0.025)
ap.add_argument("--progress", type=int, default=0)
a = ap.parse_args()
out = replay.run(a.scans, lifelong=not a.no_lifelong, mode=a.mode, period_s=a.period, progress=a.progress or None)
out.pop("poses", None); out.pop("alive_queue_index", None)
print(json.dumps(out))
