"""Pose error of the replay against the ground truth along the queue, lifelong on / off: python tools/replay_error_profile.py [n_scans]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_toolbox_amd import replay
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for lifelong in (False, True):
    q = replay.LapQueue(n)
    out = replay.run(n, lifelong=lifelong, queue=q)
    idx = np.asarray(out["alive_queue_index"])
    d = out["poses"] - q.truth[idx]
    e = np.hypot(d[:, 0], d[:, 1])
    st = out["stats"]
    print(f"lifelong {lifelong}: accepted {out['accepted']} alive {out['alive']} closures {st['loop_closures']} removed {st['nodes_removed']} "
          f"components {out['graph_components']} {out['graph_largest_components']} iou {out['map_iou_vs_truth_poses']:.3f} near {out['map_occupied_within_one_cell_of_truth_map']:.3f} rms {out['pose_error_xy_rms_m']:.3f} max {out['pose_error_xy_max_m']:.3f}")
    for lo in range(0, n, max(1, n // 10)):
        sel = (idx >= lo) & (idx < lo + n // 10)
        if sel.any():
            print(f"   queue {lo:5d}..: {int(sel.sum()):4d} alive, error rms {np.sqrt((e[sel] ** 2).mean()):.3f} max {e[sel].max():.3f}")
