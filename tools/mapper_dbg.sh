cd $GRAFT_REPO_ROOT
for inc in 1 0; do
echo "=== KH_SPA_INCREMENTAL=$inc"
KH_SPA_INCREMENTAL=$inc timeout 600 python - <<'PY'
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from slam_toolbox_amd import synth
from slam_toolbox_amd.mapper import Mapper
n_scans, loop_dist, kind = 2000, 3.0, "laps"
LIB = os.path.join("oracle", "_ref", "libkarto_ref_slam.so")
subprocess.run([sys.executable, "tests/ref_slam_runner.py", LIB, str(n_scans), str(loop_dist), "/tmp/ref", kind], check=True, timeout=900)
world = synth.make_world(12345)
truth, odom = synth.trajectory_laps(n_scans)
rng = np.random.default_rng(4)
ranges = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], rng) for i in range(n_scans)]))
m = Mapper(synth.Laser(), loop_search_maximum_distance=loop_dist, log_path="/tmp/hip.log")
for i in range(n_scans):
    m.Process(ranges[i], odom[i], 0.1 * i)
m.set_log(None); m.close()
a = open("/tmp/ref.log").read().splitlines(); b = open("/tmp/hip.log").read().splitlines()
def norm(l): return " ".join(l.split()[:2]) if l.startswith("X ") else l
k = next((i for i, (x, y) in enumerate(zip(a, b)) if norm(x) != norm(y) and not x.startswith("Z ")), None)
print("lines", len(a), len(b), "first difference at raw line", k)
if k is not None:
    for j in range(max(0, k - 6), k + 2):
        print("  ref:", a[j][:110]); print("  hip:", b[j][:110])
PY
done
