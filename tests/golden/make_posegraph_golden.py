"""Generates tests/golden/posegraph_small.{g2o,khpg,npz}: a 40-node / 75-edge synthetic pose graph written by the
CPU oracle (oracle/posegraph.py); the .npz holds the arrays the two files must parse back to, bit for bit.
Also posegraph_g2o_sample.g2o: a hand-written file in the layout g2o's own tools emit (mixed spacing, comment lines,
integer-looking numbers, no FIX record).   usage: python tests/golden/make_posegraph_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import posegraph  # noqa: E402
from slam_toolbox_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
g = synth.make_pose_graph(40, 75, seed=77)
ids = np.arange(40, dtype=np.int32) * 3 + 5            # non-contiguous unique ids
edges = ids[g["edges"]]
info = np.stack([posegraph.information_upper(c) for c in g["cov"]])
posegraph.write_text(os.path.join(HERE, "posegraph_small.g2o"), ids, g["init"], edges, g["z"], info)
posegraph.write_binary(os.path.join(HERE, "posegraph_small.khpg"), ids, g["init"], edges, g["z"], info)
np.savez(os.path.join(HERE, "posegraph_small.npz"), ids=ids, poses=g["init"], edges=edges, z=g["z"], info=info,
         cov=g["cov"])
with open(os.path.join(HERE, "posegraph_g2o_sample.g2o"), "w") as f:
    f.write("# hand-written sample in the layout of g2o's 2D datasets\n\n")
    f.write("VERTEX_SE2 0 0 0 0\n")
    f.write("VERTEX_SE2 1   1.030390 0.011350 -0.081596\r\n")
    f.write("VERTEX_SE2\t2 2.036137 -0.129733 -0.301887\n")
    f.write("  VERTEX_SE2 3 3.015097 -0.442395 -0.345514\n")
    f.write("EDGE_SE2 0 1 1.030390 0.011350 -0.081596 44.721360 0 0 44.721360 0 30.901699\n")
    f.write("EDGE_SE2 1 2 1.013900 -0.058639 -0.220291 44.721360 0.0 0.0 44.721360 0.0 30.901699\n")
    f.write("EDGE_SE2 2 3 1.027650 -0.007456 -0.043627 44.721360 0 0 44.721360 0 30.901699\n")
    f.write("# a loop closure\n")
    f.write("EDGE_SE2 3 0 -2.7e0 1.3 3.1e-1 2e1 1e0 -5e-1 25 2 1.5e1\n")
print("written")
