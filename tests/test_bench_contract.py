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
def _round_number(tag):
    return int("".join(ch for ch in tag if ch.isdigit()) or 0)


ROUND = sorted((os.path.basename(p).split("_")[0] for p in glob.glob(os.path.join(PROF, "r*_bench_line.json"))), key=_round_number)[-1]
# the dominant kernel: the windowed scoring kernel through round 3, the LDS-staged one from round 4 on
SCORE_KERNEL = "k_score<1, 8, 1" if ROUND in ("r1", "r2", "r3") else "k_score_lds<1, 4"     # (r5: k_score_lds<1, 4, true>)


def per_step_matches(d):
    return d["config"]["matches_per_step_per_gpu"] * d["n_gpus"]


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
    # "l1" / "l2" / "lds": the dominant kernel reads cache- or LDS-resident windows (report the resource that binds)
    assert r["bound"] in ("hbm", "mfma", "l1", "l2", "lds") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if ROUND != "r1":
        assert 0.0 < r["frac"] <= 1.0, "a roofline fraction is a fraction"
        assert r["hbm_frac"] is None or 0.0 < r["hbm_frac"] <= 1.0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    if _round_number(ROUND) >= 5:
        # the figures that keep their denominator from round to round: reference lookups per CU clock with and
        # without the skipped windows, the LDS array fraction, the kernels either side of the scoring kernel from the same HIP
        # events, the CPU quota beside the core count, and a timed region of at least 150 steps beside the K steps asked for
        for k in ("lookups_per_cu_clk", "lookups_per_cu_clk_read", "lookups_per_cu_clk_peak_b32", "lds_array_frac", "side_kernels_ms_per_launch"):
            assert k in r, k
        assert 0.0 < r["lookups_per_cu_clk_read"] <= r["lookups_per_cu_clk"]
        assert r["lookups_per_cu_clk_read"] < r["lookups_per_cu_clk_peak_b32"]
        assert abs(r["lds_array_frac"] - 0.5 * r["frac"]) < 1e-9
        assert set(r["side_kernels_ms_per_launch"]) == {"k_offsets_lds", "k_ties"}
        assert "cpu_quota" in c and (c["cpu_quota"] is None or 0.0 < c["cpu_quota"] <= c["cores"])
        w = d["value_windows"]
        assert w["steps"] >= 150 and w["n"] >= 5 and w["min"] <= w["median"] <= w["max"]
        assert abs(w["value"] - per_step_matches(d) * w["steps"] / w["seconds"]) / w["value"] < 1e-6
        # the fused path of one MatchScan carried the replay's sequential matches
        assert d["replay_fused_matches"] >= 1000 and d["replay_fused_fine_passes"] > 0.8 * d["replay_fused_matches"]
    # whole-job throughput and step time describe the same run
    per_step = d["config"]["matches_per_step_per_gpu"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_kernel_time_agrees_with_the_rocprof_summary():
    d = _line()
    with open(os.path.join(PROF, f"{ROUND}_bench_kernel_stats.csv")) as f:
        rows = [r for r in csv.DictReader(f) if SCORE_KERNEL in r["Name"]]
    assert len(rows) == 1
    avg_ms = float(rows[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - d["roofline"]["avg_launch_ms"]) / avg_ms < 0.05
    with open(os.path.join(PROF, f"{ROUND}_k_score_pmc.json")) as f:
        p = json.load(f)
    assert abs(p["avg_ns"] * 1e-6 - avg_ms) / avg_ms < 0.05
    # traffic = PMC HBM bytes per launch, scaled to the matches one launch of the run scored
    want = p["hbm_bytes_per_launch"] / p["matches_per_launch"] * d["roofline"]["matches_per_launch"]
    assert abs(d["roofline"]["traffic"] - want) / want < 0.02


def _recorded(name):
    """the summary of this round, or of the newest earlier round that collected one (a round re-collects the summaries of the
    kernels it changed; bench.py reads the newest the same way)"""
    suffix = name.split("_", 1)[1]
    have = sorted((p for p in glob.glob(os.path.join(PROF, "r[0-9]*_" + suffix))
                   if _round_number(os.path.basename(p).split("_")[0]) <= _round_number(ROUND)), key=lambda p: _round_number(os.path.basename(p).split("_")[0]))
    with open(have[-1]) as f:
        return json.load(f)


def _traffic(doc, prefixes, units):
    total = sum(d["hbm_bytes_per_launch"] * d["calls"] for k, d in doc["kernels"].items()
                if any(k.startswith(p) for p in prefixes) and "hbm_bytes_per_launch" in d and d.get("calls"))
    return total / units


def test_loop_and_solver_rooflines_carry_the_recorded_traffic():
    """`loop_rooflines` / `solve_rooflines` of the committed line: `traffic` = HBM bytes from the committed PMC summaries
    (tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE in separate passes, (2 * FETCH + WRITE) * 1024), per loop batch and per
    numeric factorisation; the summaries are re-collected after the line, so a few per cent of run-to-run spread are allowed."""
    if ROUND in ("r1", "r2"):
        return
    d = _line()
    loop = _recorded(f"{ROUND}_loop_pmc.json")
    per_stage = loop["kernels"].get("kseq_bin", loop["kernels"].get("k_raster_scan"))
    batches = per_stage["calls"] / 2.0
    want = _traffic(loop, ("k_raster_", "k_find_valid", "k_cell_", "k_active_set", "k_repitch", "kseq_prep", "kseq_links", "kseq_bin", "kseq_tile", "kseq_stage"), batches)
    got = d["loop_rooflines"][0]["traffic"]
    assert got is not None and abs(got - want) / want < 0.03
    assert 0.0 < d["loop_rooflines"][0]["frac"] <= 1.0
    spa = _recorded(f"{ROUND}_spa_pmc.json")
    want = _traffic(spa, ("k_potrf", "k_trsm", "k_syrk", "k_front_update", "k_extend_add"), spa["factorizations"])
    k6 = d["solve_rooflines"][0]
    assert k6["traffic"] is not None and abs(k6["traffic"] - want) / want < 0.05
    assert 0.0 < k6["frac"] <= 1.0 and k6["unit"] == "TFLOP/s"
    # flops = 2 x multiply-adds of the symbolic analysis, per factorisation
    assert abs(k6["flops_per_factorization"] - 2.0 * k6["multiply_adds_per_factorization"]) < 1.0


def test_the_printed_line_carries_the_solver_roofline():
    """Round 5's driver line lost `solve_rooflines` to the line budget: from round 6 on it is one of the ordered
    keys, and the compact line the bench prints (profiles/rN_bench_line_compact.json) must hold it, with the same numbers as the
    full record, next to every *_cpu_baseline's cores AND cpu_quota."""
    if _round_number(ROUND) < 6:
        return
    with open(os.path.join(PROF, f"{ROUND}_bench_line_compact.json")) as f:
        c = json.loads(f.read().strip().splitlines()[-1])
    d = _line()
    assert "solve_rooflines" in c and len(c["solve_rooflines"]) == 2
    for a, b in zip(c["solve_rooflines"], d["solve_rooflines"]):
        for k in ("achieved", "peak", "frac", "unit", "traffic"):
            assert k in a
            assert a[k] == b[k] or abs(a[k] - b[k]) <= 1e-5 * abs(b[k]), (k, a[k], b[k])
    for k in ("cpu_baseline", "loop_cpu_baseline", "replay_cpu_baseline"):
        if k in d:
            assert "cores" in d[k] and "cpu_quota" in d[k], k
    # the profiles the solver's figures come from are this round's
    assert os.path.exists(os.path.join(PROF, f"{ROUND}_spa_pmc.json")) and os.path.exists(os.path.join(PROF, f"{ROUND}_spa_levels_timeline.txt"))
