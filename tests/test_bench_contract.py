"""CPU: the committed bench line (profiles/r1_bench_line.json, written by `python bench.py` on the GPU box) carries
every key of the bench contract, and the kernel time it reports agrees with the committed rocprofv3 summary of the
same command (profiles/r1_bench_kernel_stats.csv) and with the PMC reduction (profiles/r1_k_score_pmc.json)."""
import csv
import json
import os

import glob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
# the newest round's line (profiles/rN_bench_line.json) with the rocprof / PMC summaries of the same round
ROUND = sorted(os.path.basename(p).split("_")[0] for p in glob.glob(os.path.join(PROF, "r*_bench_line.json")))[-1]


def _line():
    with open(os.path.join(PROF, f"{ROUND}_bench_line.json")) as f:
        return json.load(f)


def test_line_has_the_contract_keys():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "scan-matches/sec" and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # "l1" / "l2": the dominant kernel is a cache-resident gather (report the resource that binds)
    assert r["bound"] in ("hbm", "mfma", "l1", "l2") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if ROUND != "r1":
        assert 0.0 < r["frac"] <= 1.0, "a roofline fraction is a fraction"
        assert r["hbm_frac"] is None or 0.0 < r["hbm_frac"] <= 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # whole-job throughput and step time describe the same run
    per_step = d["config"]["matches_per_step_per_gpu"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_kernel_time_agrees_with_the_rocprof_summary():
    d = _line()
    with open(os.path.join(PROF, f"{ROUND}_bench_kernel_stats.csv")) as f:
        rows = [r for r in csv.DictReader(f) if "k_score<1, 8, 1>" in r["Name"]]
    assert len(rows) == 1
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05
    with open(os.path.join(PROF, f"{ROUND}_k_score_pmc.json")) as f:
        p = json.load(f)
    assert abs(p["avg_ns"] * 1e-6 - avg_ms) / avg_ms < 0.05
    # traffic = PMC HBM bytes per launch, scaled to the matches one launch of the run scored
    want = p["hbm_bytes_per_launch"] / p["matches_per_launch"] * d["roofline"]["matches_per_launch"]
    assert abs(d["roofline"]["traffic"] - want) / want < 0.02
