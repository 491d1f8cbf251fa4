"""bench.py honours --gpus: `python bench.py --gpus 2` with no launcher starts its two ranks itself (gloo: both share the one GPU of the
test box), prints ONE line with n_gpus = 2 that carries the N > 1 keys, and refuses a launcher whose world size differs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_gpus_2_spawns_its_ranks(kartohip_lib):
    env = dict(os.environ, KH_BENCH_BACKEND="gloo", KH_BENCH_WATCHDOG="900")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--details", ""], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) <= 6000
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    assert d["solve_ms_edge_sharded"] > 0
    assert d["strong_scaling"]["n_gpus"] == 2 and d["strong_scaling_in_process"]["members"] == 2
    assert d["loop_batch_ms"] > 0 and d["replay_scans_per_s"] > 0 and d["replay_poses_identical_to_one_device"] is True
    # key order: every BASELINE metric sits in front of the long dictionaries
    keys = list(d)
    assert keys.index("solve_ms_edge_sharded") < keys.index("roofline") and keys.index("loop_batch_ms") < keys.index("roofline")


@pytest.mark.gpu
def test_bench_refuses_a_mismatched_world_size(kartohip_lib):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_bench_line_builder_keeps_the_metrics_and_the_budget():
    """CPU: the line builder puts the contract's keys and every BASELINE metric first, drops the long texts and stays under its budget."""
    sys.path.insert(0, ROOT)
    import bench
    full = {"metric": "scan-matches/sec", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": {"workload": "w" * 30},
            "solve_ms": 12.0, "solve_ms_cached_analysis": 8.0, "loop_batch_ms": 9.0, "replay_scans_per_s": 3000.0,
            "roofline": {"bound": "lds", "frac": 0.3, "note": "n" * 5000}, "cpu_baseline": {"value": 50.0, "sample": "s" * 900, "forms": [1] * 50},
            "loop_workload": "x" * 4000, "zz_big": {"k%d" % i: "y" * 50 for i in range(400)}}
    line = bench.build_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_BUDGET
    keys = list(line)
    for k in ("solve_ms", "solve_ms_cached_analysis", "loop_batch_ms", "replay_scans_per_s"):
        assert keys.index(k) < keys.index("roofline")
    assert "note" not in line["roofline"] and "loop_workload" not in line and "zz_big" not in line
