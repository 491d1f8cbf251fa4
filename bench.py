"""bench.py -- headline metric of BASELINE.json on MI355X.

  python bench.py --gpus N --steps K --warmup W

metric  : scan-matches/sec (BASELINE.json "scan-matches/sec + loop-closure solve ms, 10k-node graph")
workload: BASELINE config[1] = single-scan CorrelateScan, 1081 beams, 0.3 m x 0.3 m x +-20 deg search @
          5 mm / 0.5 deg (61 x 61 x 81 = 301 401 poses x 1081 beams, 8087^2 uint8 grid), one batch of
          `--batch` independent (query scan, rasterised grid) pairs per step.  The grids (65.4 MB each)
          and every device buffer are resident in HBM before the timed region; a step is one
          kh_matcher_correlate_batch call through the C ABI: exact host tables -> K2 offsets -> K3
          scoring -> K4 ties -> tiny D2H -> host finalisation (mean + covariance).
multi-GPU: matches are independent units -> sharded across ranks, no data-path collective (weak scaling);
          one process per GPU, torch.distributed(nccl = RCCL) only for the barrier / max-over-ranks.
Also reported (extra keys): loop-closure solve ms of the 10k-node / 30k-edge SPA problem (config[3]).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

P_BEAMS = 1081
C2 = dict(nx=61, ny=61, na=81)
# B_corr = nPoses*P*(4+1) + nPoses*32 + A*P*4 bytes per CorrelateScan
ALG_BYTES_C2 = C2["nx"] * C2["ny"] * C2["na"] * P_BEAMS * 5 + C2["nx"] * C2["ny"] * C2["na"] * 32 + C2["na"] * P_BEAMS * 4
HBM_PEAK_GBS = 8000.0
# L1 (TCP) data path: 64 B per clock per CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md: chip parameters, L1 32 KiB/CU)
L1_PEAK_GBS = 64.0 * 256 * 2.4
# measured on the box (tools/tcp_ceiling.hip, profiles/r3_tcp_ceiling.txt): a loop of k_score-shaped loads (four 64-byte row
# segments per wave-level dword load) over an L1-resident window returns 37.6 TB/s = 4.2 clocks per load, whether the
# segments touch 4 or 8 lines; the same loop over a window that misses the L1 and hits the L2 needs 8.0 (4 lines) to 10.7
# (8 lines) clocks per load, i.e. about 1 clock more per missing line
L1_MEASURED_CEILING_GBS = 37600.0
L1_HIT_CLOCKS_PER_LOAD, L1_MISS_CLOCKS_PER_LINE = 4.2, 0.95
FP64_PEAK_TFLOPS = 78.6            # MI355X FP64 vector = matrix peak (spec); v_mfma_f64_16x16x4_f64
# LDS read port for ds_read_b32: 128 B per clock per CU (MI355X_MICROARCH.md, LDS table) x 256 CUs x 2.4 GHz
LDS_B32_PEAK_GBS = 128.0 * 256 * 2.4
SCORE_KERNEL = "k_score_lds<1,4,true>"


def _newest_pmc():
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_k_score_pmc.json")),
                   key=lambda p: int("".join(ch for ch in os.path.basename(p).split("_")[0] if ch.isdigit()) or 0))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r1_k_score_pmc.json")


PMC_FILE = _newest_pmc()


def round_of(path):
    """round number of profiles/rN_<name> (numeric: r10 is newer than r9)"""
    name = os.path.basename(path)
    digits = name[1:name.index("_")] if "_" in name else ""
    return int("".join(ch for ch in digits if ch.isdigit()) or 0)


def newest_profile(suffix):
    """profiles/rN_<suffix> of the newest round that has one (the summaries are re-collected in the rounds that change their kernels)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_" + suffix)), key=round_of)
    return os.path.basename(found[-1]) if found else "r3_" + suffix


def recorded_kernels(name):
    """per-kernel records of a committed rocprofv3 summary (tools/pmc_summary.py): {} when the file is missing"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def recorded_traffic(doc, kernels, units):
    """HBM bytes per unit of work: sum over the named kernels (prefix match) of recorded bytes per launch x recorded launches,
    divided by the units of work the recorded run did.  None when nothing was recorded."""
    total, seen = 0.0, False
    for k, d in doc.get("kernels", {}).items():
        if any(k.startswith(n) for n in kernels) and "hbm_bytes_per_launch" in d and d.get("calls"):
            total += d["hbm_bytes_per_launch"] * d["calls"]
            seen = True
    return total / units if seen and units else None


def pmc_rates():
    """(L2 hit rate, L1 hit rate) of k_score from the same committed PMC passes, for the roofline note."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        return 100.0 * d["l2_hit_rate"], 100.0 * d["l1_hit_rate"]
    except (OSError, KeyError, ValueError):
        return float("nan"), float("nan")


def pmc_recorded():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def pmc_traffic(batch):
    """HBM bytes per k_score launch from the committed rocprofv3 PMC passes of this same command
    (tools/prof_bench.sh + tools/pmc_traffic.py; FETCH_SIZE / WRITE_SIZE collected in separate passes and
    corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the timed run, so
    the figure is the recorded one, scaled to this run's batch when the batch differs."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        per_match = d["hbm_bytes_per_launch"] / float(d.get("matches_per_launch", 32))
        return per_match * batch
    except (OSError, KeyError, ValueError):
        return None


class _StdoutToStderr:
    """The reference's karto_sdk prints to stdout ("Registering sensor: ..."); bench.py must print ONE JSON line, so
    file descriptor 1 is pointed at stderr while the CPU baseline runs."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max, else v1 cfs quota / period); None = no limit set.  `cores` of the
    CPU baselines is what the box HAS (os.cpu_count()); this is what the process GETS."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        return None if quota <= 0 else quota / period
    except (OSError, ValueError):
        return None


def cpu_baseline(budget_s=24.0):
    """The reference's own CorrelateScan (oracle/_ref, row-parallel thread-pool stand-in for tbb::parallel_for_each) --
    or the C restatement when _ref is absent -- on the host cores, in two forms: ONE search spread over all cores (the
    latency form: only 61 rows to spread, so at most 61 cores are busy) and T concurrent matchers with cores / T
    threads each (the throughput form a batch would use).  `value` is the best throughput found; every form tried is in
    `forms`.  Bounded to about `budget_s` seconds."""
    import threading
    from common import C2_PARAMS, LASER, PRESETS, Scenario
    cores = os.cpu_count() or 1
    sc = Scenario(seed=7, n_base=10, start=0)
    args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
    from oracle import ref
    kind = "reference" if ref.available() else "port"
    if kind == "reference":
        ref.init_laser(LASER)

    def make(threads):
        if kind == "reference":
            ref.lib().ref_set_threads(threads)        # process-wide: every concurrent matcher forks this many threads
            q, base = sc.ref_scans()
            m = ref.RefMatcher(*PRESETS["C2"]["create"], C2_PARAMS)
        else:
            from oracle import karto
            q, base = sc.oracle_scans()
            m = karto.Matcher(*PRESETS["C2"]["create"], C2_PARAMS, threads=threads)
        m.add_scans(q, base)
        return m, q, base

    concurrencies = [1] + [t for t in (4, 8, 16) if cores // t >= 4]
    per_form = budget_s / len(concurrencies)
    forms = []
    for T in concurrencies:
        threads = max(1, cores // T)
        ms = [make(threads) for _ in range(T)]
        stop_at = [0.0]
        counts = [0] * T

        def work(k):
            m, q, _ = ms[k]
            while time.time() < stop_at[0]:
                m.correlate_scan(q, sc.query_pose, *args, True, False)
                counts[k] += 1
        # warm the cores up (idle vCPUs wake slowly), then count completions in a fixed window
        for phase, secs in (("warm", min(1.5, 0.2 * per_form)), ("timed", 0.8 * per_form)):
            for k in range(T):
                counts[k] = 0
            t0 = time.time()
            stop_at[0] = t0 + secs
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.time() - t0
        n = sum(counts)
        forms.append({"concurrent_matchers": T, "threads_each": threads, "searches": n, "seconds": dt,
                      "matches_per_s": n / dt if dt > 0 else 0.0})
        del ms
    best = max(forms, key=lambda f: f["matches_per_s"])
    return {"value": best["matches_per_s"], "unit": "scan-matches/s", "cores": cores, "cpu_quota": cpu_quota(), "kind": kind,
            "sample": f"config-2 CorrelateScan (61x61x81 poses x 1081 beams) for ~{budget_s:.0f} s: best of "
                      f"{[f['concurrent_matchers'] for f in forms]} concurrent matchers = {best['concurrent_matchers']} x "
                      f"{best['threads_each']} threads; single search over all {cores} cores: {forms[0]['matches_per_s']:.1f}/s",
            "forms": forms}


def solver_leg(device=0, rank=0, world=1, cpu=True):
    """Loop-closure solve of BASELINE config[3]: 10k nodes / 30k edges (extra keys).  world > 1: every rank
    holds the graph, the linearisation is sharded by edge blocks and H, g are summed with one RCCL
    all-reduce per LM iteration (honest sizing says this is a slowdown at 30k
    edges -- it is measured, not assumed); factorisation and LM control are replicated."""
    try:
        from slam_toolbox_amd import synth
        from slam_toolbox_amd.scan_solver import HipSpaSolver
    except ImportError:
        return None
    g = synth.make_pose_graph(10000, 30000, seed=12345)
    sol = HipSpaSolver(device=device)
    collective = "none"
    if world > 1:
        # the all-reduce of H || g runs INSIDE the library (ncclAllReduce of RCCL, kh_spa_set_comm); the 128-byte
        # communicator id travels over the process group that already exists.  Should RCCL refuse (e.g. ranks sharing a
        # device in a functional run), the C-ABI callback path through torch.distributed takes over.
        import torch
        import torch.distributed as dist
        from slam_toolbox_amd import comm as khcomm
        on_gpu = dist.get_backend() == "nccl"
        uid = torch.zeros(khcomm.ID_BYTES, dtype=torch.uint8)
        why = ""
        try:
            if rank == 0:
                uid = torch.from_numpy(khcomm.unique_id().copy())
        except Exception as exc:
            why = type(exc).__name__
        uid = uid.to("cuda") if on_gpu else uid
        dist.broadcast(uid, src=0)
        # ncclCommInitRank is collective: it runs on a helper thread with a deadline, and the ranks then AGREE (min over
        # ranks) on whether everybody has a communicator -- one rank falling back alone would leave the others waiting
        # inside an all-reduce for ever
        box = {}

        def init():
            try:
                box["comm"] = khcomm.Communicator(device, rank, world, uid.cpu().numpy())
            except Exception as exc:
                box["why"] = type(exc).__name__
        import threading
        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(timeout=120.0)
        ok = torch.tensor([1 if "comm" in box else 0], dtype=torch.int32)
        ok = ok.to("cuda") if on_gpu else ok
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            sol.SetCommunicator(box["comm"])
            collective = "in-library ncclAllReduce (RCCL)"
        else:
            why = box.get("why", why or ("timeout" if th.is_alive() else "refused on another rank"))
            sol.enable_sharding(rank, world)
            collective = f"torch.distributed.all_reduce through the C-ABI callback (in-library communicator: {why})"
    # config[3] says "serialized pose graph": the graph goes through the library's own file format
    # (kh_spa_save / kh_spa_load, binary) before every solve, like loadSerializedPoseGraph rebuilds the plugin
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        # two files of the SAME graph with the extra edges in a different order: loading them in turn makes every solve a
        # cold one (the library keeps its ordering / symbolic factorisation when a reloaded graph has the topology it
        # analysed last, which is not what a mapper whose graph grew since the last closure sees)
        paths = []
        n_odo = 10000 - 1
        for k, order in enumerate((np.arange(len(g["edges"])), np.concatenate([np.arange(n_odo), np.arange(len(g["edges"]) - 1, n_odo - 1, -1)]))):
            sol.load(g["init"], g["edges"][order], g["z"][order], g["cov"][order])
            paths.append(os.path.join(tmp, f"config4_rank{rank}_{k}.khpg"))
            sol.save_graph(paths[-1], binary=True)
        sol.load_graph(paths[0])
        sol.Compute()                       # warm-up (allocation)
        times, loads, summs = [], [], []
        for rep in range(6):
            t = time.time()
            sol.load_graph(paths[(rep + 1) % 2])
            loads.append(time.time() - t)
            t = time.time()
            summs.append(dict(sol.Compute()))
            times.append(time.time() - t)
        cached = []
        for rep in range(3):                # the same file again: cached analysis
            sol.load_graph(paths[0])
            t = time.time()
            last = dict(sol.Compute())
            if rep:
                cached.append(time.time() - t)
        # GPU time per phase: one more solve with HIP events around the phases (kh_spa_set_debug bit 1; the timed solves above run
        # without them -- an event record is a 5-6 us bubble on the stream)
        sol.set_debug(phase_timing=True)
        sol.load_graph(paths[0])
        phases = dict(sol.Compute())
        sol.set_debug()
    summ = summs[int(np.argsort(times)[len(times) // 2])]        # the median run's own summary
    for key_ms in ("factor_gpu_ms", "backward_gpu_ms", "linearize_gpu_ms"):
        summ[key_ms] = phases[key_ms]
    key = "solve_ms" if world == 1 else "solve_ms_edge_sharded"
    out = {key: float(np.median(times)) * 1e3, "solve_graph_load_ms": float(np.median(loads)) * 1e3,
           "solve_iterations": int(summ["iterations"]),
           "solve_final_cost": float(summ["final_cost"]), "solve_graph": "10000 nodes / 30000 edges",
           "solve_parallelism": "1 GPU" if world == 1 else f"{world} GPUs: edge-block linearisation + all-reduce(H, g) [{collective}], replicated factorisation",
           "solve_symbolic_ms": float(summ["symbolic_ms"]), "solve_nnz_factor": int(summ["nnz_factor"]),
           "solve_ms_cached_analysis": float(np.median(cached)) * 1e3 if cached else None,
           "solve_levels": int(summ["levels"])}
    if world == 1 and summ["factorizations"] > 0:
        # K6: flops of the numeric factorisations / GPU time of the assemble + factor + forward sweeps (HIP events)
        # factor_flops counts multiply-adds (sum over the fronts of sum_j (m - j)^2): two floating-point operations each
        flops = 2.0 * float(summ["factor_flops"]) * summ["factorizations"]
        tf = flops / (summ["factor_gpu_ms"] * 1e-3) / 1e12 if summ["factor_gpu_ms"] > 0 else 0.0
        n_lin = summ["successful_steps"]          # evaluation points linearised (the start + every accepted step)
        lin_bytes = 480.0 * 30000 * n_lin         # 144 B read + 336 B written per edge
        lin_gbs = lin_bytes / (summ["linearize_gpu_ms"] * 1e-3) / 1e9 if summ["linearize_gpu_ms"] > 0 else 0.0
        spa_name = newest_profile("spa_pmc.json")
        spa_doc = recorded_kernels(spa_name)
        k6_traffic = recorded_traffic(spa_doc, ("k_potrf", "k_trsm", "k_syrk", "k_front_update", "k_extend_add"), spa_doc.get("factorizations"))
        n_lin_rec = sum(spa_doc.get("kernels", {}).get(k, {}).get("calls", 0) for k in ("k_gather_H", "k_gather_Hg_norms"))
        k5_traffic = recorded_traffic(spa_doc, ("k_edge_lin", "k_gather_H", "k_gather_g"), n_lin_rec)
        out["solve_rooflines"] = [
            {"kernel": "K6 level pipeline: k_potrf + k_trsm + k_syrk + k_front_update (+ assemble), multifrontal Cholesky with the forward solve fused",
             "bound": "latency (dependent levels: the pivot chains of k_potrf); ceiling quoted = mfma f64",
             "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS,
             "flops_per_factorization": 2.0 * float(summ["factor_flops"]), "multiply_adds_per_factorization": float(summ["factor_flops"]),
             "factorizations": int(summ["factorizations"]),
             "gpu_ms": float(summ["factor_gpu_ms"]), "levels": int(summ["levels"]), "nnz_factor": int(summ["nnz_factor"]),
             "traffic": k6_traffic, "traffic_source": "recorded: profiles/" + spa_name + " (rocprofv3 PMC passes of tools/quick_spa.py, the same "
                                                      "graph): HBM bytes of k_potrf + k_trsm + k_syrk + k_front_update per numeric factorisation",
             "timing": "HIP events around the phases in one extra solve (kh_spa_set_debug bit 1); the timed solves run without them"},
            {"kernel": "K5 k_edge_lin + k_gather_H / _g (normal equations)", "bound": "hbm", "achieved": lin_gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": lin_gbs / HBM_PEAK_GBS, "algorithmic_bytes": lin_bytes, "linearizations": int(n_lin),
             "gpu_ms": float(summ["linearize_gpu_ms"]), "traffic": k5_traffic,
             "traffic_source": "recorded: profiles/" + spa_name + ": HBM bytes of k_edge_lin + k_gather_H + k_gather_g per linearisation",
             "note": "480 B per edge algorithmic (SURVEY 8d); 14.4 MB per linearisation is latency-, not bandwidth-sized"},
        ]
        out["solve_backward_gpu_ms"] = float(summ["backward_gpu_ms"])
    if world == 1 and cpu:
        # CPU beside it: Ceres is not available, so the timed thing is the numpy / scipy restatement
        try:
            from oracle import spa
            t = time.time()
            _, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
            out["solve_cpu_baseline"] = {"value": (time.time() - t) * 1e3, "unit": "ms", "cores": 1, "cpu_quota": cpu_quota(), "kind": "port",
                                         "sample": f"oracle/spa.py (numpy + scipy SuperLU restatement of the Ceres LM, not Ceres itself), "
                                                   f"one solve of the same graph, {info['iterations']} iterations"}
        except Exception as exc:
            out["solve_cpu_baseline"] = {"error": repr(exc)[:200]}
    sol.close()
    return out


def loop_leg(device=0, n_pairs=256, batch=256, cpu=True, resident=True):
    """BASELINE config[2]: loop-closure candidate batch -- 256 DISTINCT (query scan, candidate chain) pairs on the
    2k-node trajectory (synth.loop_batch: chain length 10-40 scans); each pair = preset L coarse MatchScan
    (doPenalize=False, doRefineMatch=False, Mapper.cpp:1511-1512) and, for those passing the coarse gate (response >
    0.35, both variances < 9.0, offline.yaml:43-45), a preset S coarse+fine MatchScan (Mapper.cpp:1533-1535).  Unit of
    work = one pair (extra keys, rank 0).  tests/test_baseline_shapes_gpu.py checks this very batch against the oracle."""
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcher
    lb = synth.loop_batch(n_pairs)
    cache = {}

    def scan_at(i):
        # the graph's scans live in HBM (kh_scan.device_points_xy): inputs resident on the device before the timed region,
        # like the mapper front end keeps them (csrc/mapper_host.cpp); loop_batch_ms_host_scans is the same batch with
        # the base scans uploaded by every call
        if i not in cache:
            cache[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res)
            if resident:
                cache[i].MakeResident(device)
        return cache[i]
    queries = [LocalizedRangeScan(lb["ranges"][q], pose, LASER.min_angle, LASER.ang_res) for q, pose, _ in lb["pairs"]]
    chains = [[scan_at(i) for i in chain] for _, _, chain in lb["pairs"]]
    mp = MapperParams(**OFFLINE_PARAMS)
    mL = ScanMatcher.Create(mp, *PRESETS["L"]["create"], device=device, max_batch=batch)
    mS = ScanMatcher.Create(mp, *PRESETS["S"]["create"], device=device, max_batch=batch)

    # kh_scan arrays are marshalled once per distinct batch composition (what a C++ caller has for free)
    packs = {}

    def packed(ids):
        key = tuple(ids)
        if key not in packs:
            packs[key] = ScanMatcher.pack_batch([queries[i] for i in ids], [chains[i] for i in ids])
        return packs[key]

    from slam_toolbox_amd.scan_matcher import LoopClosureBatch

    def run(pieces=1):
        # MapperGraph::TryCloseLoop's two matches per chain (Mapper.cpp:1515-1549) through kh_loop_closure_batch: preset L coarse
        # match -> gate -> preset S match of the temporary scan at the coarse pose, the batch cut into `pieces` so that the two
        # matchers (two handles, two streams) can overlap; pieces = 1: the two stages back to back, which is what measures
        # fastest (`loop_batch_ms_four_pieces` beside it: the stages of neighbouring pieces compete for the same host pool and
        # the same GPU, and every piece pays the fixed costs of a call again)
        table = []
        for b in range(0, n_pairs, batch):
            ids = list(range(b, min(n_pairs, b + batch)))
            o = LoopClosureBatch(mL, mS, None, None, LASER.min_angle, LASER.ang_res, 0.35, 9.0, pieces=pieces, packed=packed(ids))
            table.append((len(ids), int(o["passed"].sum())))
        return table
    run()                                                     # warm-up: allocations
    times = []
    for _ in range(5):
        t = time.perf_counter()
        table = run()
        times.append(time.perf_counter() - t)
    serial = []
    run(pieces=4)
    for _ in range(3):
        t = time.perf_counter()
        run(pieces=4)
        serial.append(time.perf_counter() - t)
    # one more pass with the library's event timers on: GPU time of the rasteriser (K1) and of the scoring kernel (the
    # event pairs synchronise the host: unpipelined, so that the four figures do not contain each other's kernels)
    mL.profile(True); mS.profile(True)
    run(pieces=1)
    pL, pS = mL.profile(False), mS.profile(False)
    n_ok = sum(t[1] for t in table)
    mL.close(); mS.close()
    med = float(np.median(times))
    out = {"loop_pairs_per_s": n_pairs / med, "loop_batch_ms": med * 1e3, "loop_batch_ms_four_pieces": float(np.median(serial)) * 1e3,
           "loop_workload": f"{n_pairs} distinct pairs (chains 10-40 scans): preset L coarse MatchScan, "
                            f"{n_ok} of them passing the gate -> preset S coarse+fine match of the temporary scan at the coarse pose "
                            f"(kh_loop_closure_batch, one piece)",
           "loop_gpu_ms": {"raster_L": pL["raster_ms"], "score_L": pL["score_ms"], "raster_S": pS["raster_ms"], "score_S": pS["score_ms"]}}
    # K1 roofline (HBM): SURVEY 8d B_rast = grid bytes (clear) + 16 B per point + 2 k^2 per new cell; reported against the
    # grid bytes + points, the part that is compulsory for any implementation that clears the grid
    pts = sum(len(c) for c in chains) * P_BEAMS
    gridL, gridS = 968 * 965, 4096 * 4093
    k1_bytes = n_pairs * gridL + n_ok * gridS + 16.0 * pts * (1 + n_ok / max(1, n_pairs))
    k1_ms = pL["raster_ms"] + pS["raster_ms"]
    if k1_ms > 0:
        gbs = k1_bytes / (k1_ms * 1e-3) / 1e9
        # (round 5: the first-point rasteriser kseq_prep_batch / kseq_links / kseq_bin / kseq_tile / kseq_stage carries the batches of
        # both presets; the hash-table passes k_cell_* / k_active_set / k_raster_bin remain for handles whose tables do not fit)
        K1 = ("k_raster_", "k_find_valid", "k_cell_", "k_active_set", "k_repitch", "kseq_prep", "kseq_links", "kseq_bin", "kseq_tile", "kseq_stage")
        loop_name = newest_profile("loop_pmc.json")
        doc = recorded_kernels(loop_name)
        per_stage = doc.get("kernels", {}).get("kseq_bin", doc.get("kernels", {}).get("k_raster_scan", {}))
        batches = per_stage.get("calls", 0) / 2.0                                                # one launch per stage (L, S) of a batch
        traffic = recorded_traffic(doc, K1, batches)
        out["loop_rooflines"] = [{"kernel": "K1 rasteriser (kseq_prep_batch + kseq_links + kseq_bin + kseq_tile / k_raster_tile + k_repitch*), presets L and S", "bound": "hbm",
                                  "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": k1_bytes,
                                  "gpu_ms": k1_ms, "traffic": traffic,
                                  "traffic_gbs": (traffic / (k1_ms * 1e-3) / 1e9) if traffic else None,
                                  "traffic_source": "recorded: profiles/" + loop_name + " (rocprofv3 PMC passes of tools/loop_pieces.py, the same 256-pair "
                                                    "batch; (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the K1 kernels, per batch)",
                                  "note": "algorithmic_bytes still prices Grid::Clear at the whole grid (SURVEY 8d); the kernels zero only the tiles "
                                          "the slot's previous rasterisation wrote, so the measured traffic is BELOW the algorithmic figure for the "
                                          "clear and above it for the cell-table and tile-list passes"}]
    if resident:
        host = loop_leg(device, n_pairs, batch, cpu=False, resident=False)
        out["loop_batch_ms_host_scans"] = host["loop_batch_ms"]
        out["loop_scan_store"] = ("base scans resident in HBM (kh_scan.device_points_xy, 17.3 KB per scan); loop_batch_ms_host_scans: "
                                  "the same batch with the ~1600 distinct base scans (27.8 MB) gathered and uploaded by each of the two calls")
    if cpu:
        out["loop_cpu_baseline"] = loop_cpu_baseline(lb, n_sample=32)
    return out


def loop_cpu_baseline(lb, n_sample=32):
    """The reference's own MatchScan (oracle/_ref) -- preset L coarse + preset S coarse+fine for the pairs passing the
    gate -- on a sample of the loop batch, pairs dealt to T concurrent matcher pairs with cores / T threads each."""
    import threading
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from oracle import ref
    cores = os.cpu_count() or 1
    if not ref.available():
        return {"error": "oracle/_ref not built"}
    with _StdoutToStderr():
        ref.init_laser(LASER)
        T = max(1, min(8, cores // 8))
        ref.lib().ref_set_threads(max(1, cores // T))
        sample = lb["pairs"][:: max(1, len(lb["pairs"]) // n_sample)][:n_sample]
        scans = {}

        def rscan(i):
            if i not in scans:
                scans[i] = ref.RefScan(lb["ranges"][i], lb["truth"][i])
            return scans[i]
        work = [(ref.RefScan(lb["ranges"][q], pose), [rscan(i) for i in chain]) for q, pose, chain in sample]
        matchers = [(ref.RefMatcher(*PRESETS["L"]["create"], OFFLINE_PARAMS), ref.RefMatcher(*PRESETS["S"]["create"], OFFLINE_PARAMS))
                    for _ in range(T)]
        nxt = [0]
        lock = threading.Lock()

        def worker(k):
            mL, mS = matchers[k]
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= len(work):
                    return
                q, base = work[i]
                r, _, c = mL.match_scan(q, base, False, False)
                if r > 0.35 and c[0, 0] < 9.0 and c[1, 1] < 9.0:
                    mS.match_scan(q, base, False, True)
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.time() - t0
    return {"value": len(work) / dt, "unit": "pairs/s", "cores": cores, "cpu_quota": cpu_quota(), "kind": "reference",
            "sample": f"{len(work)} of the batch's pairs, {T} concurrent (L, S) matcher pairs x {max(1, cores // T)} threads, {dt:.1f} s"}


def strong_leg(device=0, rank=0, world=1, n_pairs=2048, distinct=256, batch=256):
    """Strong scaling of the candidate matcher: a FIXED set of n_pairs loop-closure candidate
    pairs (the config[2] batch, tiled) is dealt round-robin to the ranks, every rank coarse-matches its share on its GPU
    (preset L MatchScan, doPenalize / doRefineMatch false), the 13 result doubles per pair are collected with one fixed-size
    all-gather (RCCL with backend nccl) and TryCloseLoop's first-acceptance rule is applied to the gathered table.  Time =
    max over ranks from the first match to the gathered table.  Every rank takes part; rank 0 reports (extra key)."""
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from slam_toolbox_amd import shard, synth
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcher
    import torch
    lb = synth.loop_batch(distinct)
    cache = {}

    def scan_at(i):
        if i not in cache:
            cache[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res).MakeResident(device)
        return cache[i]
    queries = [LocalizedRangeScan(lb["ranges"][q], pose, LASER.min_angle, LASER.ang_res) for q, pose, _ in lb["pairs"]]
    chains = [[scan_at(i) for i in chain] for _, _, chain in lb["pairs"]]
    m = ScanMatcher.Create(MapperParams(**OFFLINE_PARAMS), *PRESETS["L"]["create"], device=device, max_batch=batch)
    packs = {}

    def match_fn(units):
        key = tuple(u % distinct for u in units)
        if key not in packs:
            packs[key] = ScanMatcher.pack_batch([queries[u] for u in key], [chains[u] for u in key])
        resp, means, covs, _ = m.MatchScanBatch(None, None, False, False, packed=packs[key])
        return resp, means, covs
    dev = "cuda" if world > 1 else "cpu"

    def once():
        return shard.match_candidates_sharded(match_fn, n_pairs, rank, world, batch=batch, device=dev)
    once()                                                    # warm-up: allocations, marshalling
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        t = time.perf_counter()
        table = once()
        times.append(shard.max_over_ranks(time.perf_counter() - t, device=dev))
    m.close()
    med = float(np.median(times))
    return {"strong_scaling": {"pairs": n_pairs, "n_gpus": world, "ms": med * 1e3, "pairs_per_s": n_pairs / med,
                               "first_accepted": int(shard.first_accepted(table, 0.35, 9.0)),
                               "workload": f"{n_pairs} candidate pairs ({distinct} distinct, tiled) preset L coarse MatchScan, "
                                           f"round-robin over {world} rank(s), one all-gather of {n_pairs} x 13 doubles"}}


def latency_leg(device=0):
    """One call at a time -- what karto::ScanMatcher's interface gives slam_toolbox (include/karto_hip/karto_adaptor.hpp:
    HipScanMatcher packs and uploads the base scans with every MatchScan) next to the same call with the base scans
    resident in HBM (kh_scan.device_points_xy, what kh_mapper does).  Presets: S sequential (10 base scans), L loop (20),
    C2 = BASELINE config[1]'s geometry as a full MatchScan (coarse 31 x 31 x 81 poses at 1 cm / 0.5 deg, then the fine
    pass) and as the single CorrelateScan over 61 x 61 x 81 poses the headline batches up."""
    import math
    from common import Scenario, make_hip_matcher
    out = {}
    for preset, nbase in (("S", 10), ("L", 20), ("C2", 10)):
        sc = Scenario(seed=11, n_base=nbase, start=20)
        q, b = sc.hip_scans()
        hm = make_hip_matcher(preset)
        row = {}
        for label in ("base_scans_uploaded_per_call", "base_scans_resident"):
            if label == "base_scans_resident":
                for scan in b:
                    scan.MakeResident(device)
            for _ in range(5):
                hm.MatchScan(q, b, True, True)
            n = 100
            t = time.perf_counter()
            for _ in range(n):
                hm.MatchScan(q, b, True, True)
            row[label + "_ms"] = (time.perf_counter() - t) / n * 1e3
        row["fused_path"] = hm.seq_stats()
        if preset == "C2":
            hm.AddScans(q, b, slot=0)
            args = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))
            for _ in range(3):
                hm.CorrelateScan(q, sc.query_pose, *args, True)
            n = 30
            t = time.perf_counter()
            for _ in range(n):
                hm.CorrelateScan(q, sc.query_pose, *args, True)
            row["single_correlate_scan_ms"] = (time.perf_counter() - t) / n * 1e3
            row["single_correlate_scan_per_s"] = 1e3 / row["single_correlate_scan_ms"]
        # algorithmic bytes of ONE MatchScan = B_rast + sum of B_corr over its CorrelateScan calls (coarse, then fine), with
        # B_corr = nPoses * P * 5 + nPoses * 32 + A * P * 4 and B_rast = grid bytes (Grid::Clear) + 16 B per point -- the stamps'
        # read-modify-write term (<= cells * k^2 * 2) is left out, so the figure is a lower bound of the reference's access stream
        from common import PRESETS as _PRESETS
        size, res = _PRESETS[preset]["create"][0], _PRESETS[preset]["create"][1]
        prm = _PRESETS[preset]["params"]
        P = 1081
        nxy = int(round(size / (2.0 * res))) + 1
        na = int(round(2.0 * prm["coarse_search_angle_offset"] / prm["coarse_angle_resolution"])) + 1
        naf = int(round(prm["coarse_angle_resolution"] / prm["fine_search_angle_offset"])) + 1
        b_corr = lambda n_poses, n_ang: n_poses * P * 5 + n_poses * 32 + n_ang * P * 4
        gi = hm.grid_info()
        b_rast = int(gi["data_size"]) + sum(int(scan.points.size // 2) for scan in b) * 16
        total = b_rast + b_corr(nxy * nxy * na, na) + b_corr(9 * naf, naf)
        wall = row["base_scans_resident_ms"] * 1e-3
        row["roofline"] = {"bound": "hbm", "achieved": total / wall / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": total / wall / 1e9 / HBM_PEAK_GBS,
                           "algorithmic_bytes": total, "b_rast": b_rast, "b_corr_coarse": b_corr(nxy * nxy * na, na), "b_corr_fine": b_corr(9 * naf, naf),
                           "coarse_poses": nxy * nxy * na, "fine_poses": 9 * naf, "wall_ms": wall * 1e3, "traffic": None,
                           "note": "wall time of one call (base scans resident), nine dependent launches; kernel-time sums: profiles/rN_seq_*_kernel_stats.csv"}
        hm.close()
        out[preset] = row
    # config[1]'s geometry as FULL MatchScans in one batch (rasterise + coarse 31 x 31 x 81 + fine per pair): 64 pairs per call
    from slam_toolbox_amd.scan_matcher import ScanMatcher
    scs = [Scenario(seed=40 + i, n_base=10, start=31 * i + 7) for i in range(8)]
    distinct = [sc.hip_scans() for sc in scs]
    for _, base in distinct:
        for scan in base:
            scan.MakeResident(device)
    pairs = [distinct[i % 8] for i in range(64)]
    hm = make_hip_matcher("C2", max_batch=64)
    pack = ScanMatcher.pack_batch([q for q, _ in pairs], [b for _, b in pairs])
    for _ in range(2):
        hm.MatchScanBatch(None, None, True, True, packed=pack)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        hm.MatchScanBatch(None, None, True, True, packed=pack)
        ts.append(time.perf_counter() - t)
    hm.close()
    batch_ms = float(np.median(ts)) * 1e3
    return {"single_call_latency": out,
            # the same, short enough to sit in front of the line: ONE MatchScan (penalise + refine) per call, ms
            "match_scan_ms": {p: out[p]["base_scans_resident_ms"] for p in out} | {p + "_uploaded": out[p]["base_scans_uploaded_per_call_ms"] for p in out},
            "match_scan_roofline": {p: {k: out[p]["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes", "wall_ms", "traffic")} | {"bound": "hbm"}
                                    for p in out},
            "config2_match_scan_batch_ms": batch_ms, "config2_match_scans_per_s": 64.0 / (batch_ms * 1e-3),
            "config2_match_scan_workload": "64 (query, 10 base scans) pairs per kh_matcher_match_batch at config[1]'s geometry: AddScans + coarse "
                                           "CorrelateScan 31 x 31 x 81 poses + fine pass (SURVEY 8d), base scans resident",
            "single_call_latency_note": "wall time per call from Python through the C ABI, MatchScan(doPenalize, doRefineMatch) = true; "
                                        "C2.single_correlate_scan_ms = config[1] issued one CorrelateScan at a time"}


def group_leg(devices, n_pairs=2048, distinct=256, batch=256):
    """The same fixed set of candidate pairs as strong_leg through the IN-PROCESS multi-device batch of the library
    (kh_matcher_group_match_batch: one member per entry of `devices`, candidate i on member i % members, a host thread per
    member, results in candidate order; no collective).  This is the path kh_mapper's TryCloseLoop uses
    (kh_mapper_create_on_devices).  Rank 0 only; the graph's scans are resident on every member's device."""
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from slam_toolbox_amd import shard, synth
    from slam_toolbox_amd.comm import DeviceBuffer
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcherGroup
    lb = synth.loop_batch(distinct)
    scans, copies = {}, {}

    def scan_at(i):
        if i not in scans:
            scans[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res)
        return scans[i]

    def copy_on(i, dev):
        if (i, dev) not in copies:
            buf = DeviceBuffer(scans[i].points.size, dev)
            buf.upload(scans[i].points)
            copies[(i, dev)] = buf
        return copies[(i, dev)].ptr
    queries = [LocalizedRangeScan(lb["ranges"][q], pose, LASER.min_angle, LASER.ang_res) for q, pose, _ in lb["pairs"]]
    g = ScanMatcherGroup(MapperParams(**OFFLINE_PARAMS), *PRESETS["L"]["create"], devices=devices, max_batch_per_member=batch)
    nm = len(devices)
    units = [u % distinct for u in range(n_pairs)]
    q_list = [queries[u] for u in units]
    b_list = [[scan_at(i) for i in lb["pairs"][u][2]] for u in units]
    rows = []
    for k, u in enumerate(units):
        for i in lb["pairs"][u][2]:
            row = [0] * nm
            row[k % nm] = copy_on(i, devices[k % nm])          # only the member that reads the scan needs the copy
            rows.append(row)
    packed = g.pack_batch(q_list, b_list, np.asarray(rows, dtype=np.uint64))
    g.MatchScanBatch(None, None, False, False, packed=packed)      # warm-up: allocations
    times = []
    for _ in range(3):
        t = time.perf_counter()
        resp, means, covs, st = g.MatchScanBatch(None, None, False, False, packed=packed)
        times.append(time.perf_counter() - t)
    g.close()
    table = np.concatenate([resp[:, None], means, covs.reshape(-1, 9)], axis=1)
    med = float(np.median(times))
    return {"pairs": n_pairs, "members": nm, "devices": list(map(int, devices)), "ms": med * 1e3, "pairs_per_s": n_pairs / med,
            "first_accepted": int(shard.first_accepted(table, 0.35, 9.0)),
            "workload": f"{n_pairs} candidate pairs ({distinct} distinct, tiled) preset L coarse MatchScan through "
                        f"kh_matcher_group_match_batch, one process, {nm} member(s), no collective"}


def replay_cpu_baseline(n_scans=1500):
    """CPU baseline of config[4] AND the pin of the replay: the first n_scans of the replay queue, non-lifelong (the
    reference's lifelong node is a ROS node and cannot be built here), through the REFERENCE karto::Mapper built from
    /root/reference (oracle/_ref/libkarto_ref_slam.so: reference Mapper.cpp + reference CPU ScanMatcher on 64 host threads;
    its solver plugin is this library's, attached through karto::ScanSolver -- Ceres is not available) and through the
    library's mapper.  Reports the reference's rate, whether the two runs end with the same poses bit for bit, and the
    intersection-over-union of the two occupancy maps."""
    import ctypes as C
    from slam_toolbox_amd import replay
    from slam_toolbox_amd.mapper import Mapper
    from slam_toolbox_amd.occupancy_grid import OccupancyGrid
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    path = os.path.join(ROOT, "oracle", "_ref", "libkarto_ref_slam.so")
    if not os.path.exists(path):
        return {"replay_cpu_baseline": None, "replay_cpu_baseline_note": "oracle/_ref/libkarto_ref_slam.so not built"}
    q = replay.LapQueue(n_scans)
    ranges = np.ascontiguousarray(np.stack([q.ranges(i) for i in range(n_scans)]))
    odom = np.ascontiguousarray(q.odom)
    L = q.laser
    m = Mapper(L)
    t0 = time.perf_counter()
    ids = [i for i in range(n_scans) if m.Process(ranges[i], odom[i], 0.1 * i)[0]]
    t_hip = time.perf_counter() - t0
    poses = m.poses()
    m.close()
    lib = C.CDLL(path)
    lib.ref_init_laser.restype = C.c_int
    lib.ref_init_laser.argtypes = [C.c_double] * 6
    lib.ref_slam_run.restype = C.c_int
    lib.ref_slam_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_char_p, C.c_void_p, C.c_int]
    nb = lib.ref_init_laser(L.min_angle, L.max_angle, L.ang_res, L.min_range, L.max_range, L.range_threshold)
    threads = min(64, os.cpu_count() or 1)
    lib.ref_set_threads(threads)
    out = np.zeros((n_scans, 4))
    t0 = time.perf_counter()
    acc = lib.ref_slam_run(n_scans, nb, ranges.ctypes.data, odom.ctypes.data, 3.0, b"/tmp/kh_bench_ref_replay.log", out.ctypes.data, n_scans)
    t_ref = time.perf_counter() - t0
    ref = out[:max(acc, 0), 1:]
    same = bool(acc == len(ids) and np.array_equal(ref, poses))
    iou = None
    if acc == len(ids) and acc > 0:
        a_scans = [LocalizedRangeScan(ranges[i], p, L.min_angle, L.ang_res) for i, p in zip(ids, poses)]
        b_scans = [LocalizedRangeScan(ranges[i], p, L.min_angle, L.ang_res) for i, p in zip(ids, ref)]
        g = OccupancyGrid.CreateFromScans(a_scans + b_scans, 0.05, L, 0)
        g.Clear(); g.AddScans(a_scans, L); g.Update()
        ma = g.cells()[:, :g.width] == 100
        g.Clear(); g.AddScans(b_scans, L); g.Update()
        mb = g.cells()[:, :g.width] == 100
        iou = float((ma & mb).sum()) / max(1, int((ma | mb).sum()))
        t_scans = [LocalizedRangeScan(ranges[i], q.truth[i], L.min_angle, L.ang_res) for i in ids]
        g.Clear(); g.AddScans(t_scans, L); g.Update()
        mt = g.cells()[:, :g.width] == 100
        iou_truth = float((ma & mt).sum()) / max(1, int((ma | mt).sum()))
        g.close()
    else:
        iou_truth = None
    truth = q.truth[np.asarray(ids)]
    d = poses - truth
    return {"replay_cpu_baseline": {"value": n_scans / t_ref, "unit": "scans/s", "cores": threads, "cpu_quota": cpu_quota(), "kind": "reference",
                                    "sample": f"first {n_scans} scans of the replay queue, non-lifelong, reference karto::Mapper::Process with its "
                                              f"CPU scan matcher ({t_ref:.1f} s); the library's mapper on the same scans: {t_hip:.2f} s"},
            "replay_poses_identical_to_reference": same, "replay_map_iou_vs_reference": iou,
            "replay_nonlifelong_map_iou_vs_truth_poses": iou_truth,
            "replay_reference_scans": n_scans, "replay_reference_accepted": int(acc),
            "replay_reference_pose_error_xy_rms_m": float(np.sqrt((d[:, :2] ** 2).sum(1).mean()))}


def replay_leg(device=0, n_scans=3000, cpu=True, full_length=True):
    """BASELINE config[4]: lifelong-mode replay, end to end on one GPU -- scan queue -> mapper front end of the library
    (sequential match, links, speculative loop closure, SPA solves, node decay) -> occupancy grid (extra keys).  The
    default run replays a bounded prefix, once with every scan processed (sync) and once behind the asynchronous node's
    depth-1 queue with scans arriving every 25 ms (src/slam_toolbox_async.cpp:34-57); tools/replay.py --scans 50000 is the
    full-length run (profiles/)."""
    from slam_toolbox_amd import replay
    out = replay.run(n_scans, lifelong=True, mode="sync", device=device)
    st = out["stats"]
    res = {"replay_scans_per_s": out["scans_per_s"], "replay_workload": f"{n_scans}-scan lap circuit (odometry noise 1 % / 0.2 deg per m), "
           f"lifelong mode, sync queue: {out['accepted']} accepted, {out['alive']} alive after node decay, {st['loop_closures']} loop closures, "
           f"{st['matches']} matches", "replay_wall_s": out["wall_s"], "replay_closures": int(st["loop_closures"]),
           "replay_matches": int(st["matches"]), "replay_fused_matches": int(st.get("fused_matches", 0)),
           "replay_fused_fine_passes": int(st.get("fused_fine_passes", 0)),
           "replay_ms_split": out["ms_split"],
           "replay_map_build_ms": out["map_build_ms"], "replay_map_iou_vs_truth_poses": out["map_iou_vs_truth_poses"],
           "replay_map_occupied_within_one_cell_of_truth_map": out["map_occupied_within_one_cell_of_truth_map"],
           "replay_pose_error_xy_rms_m": out["pose_error_xy_rms_m"], "replay_pose_error_xy_max_m": out["pose_error_xy_max_m"],
           "replay_graph_components": out["graph_components"],
           "replay_lifelong_note": "the node-decay policy (slam_toolbox_lifelong.cpp:149-178, restated and pinned in tests/test_lifelong_policy_gpu.py) "
                                   "removes vertices together with their edges and keeps nothing connected: on a circuit driven lap after lap the pose "
                                   "graph falls apart into replay_graph_components pieces and the map drifts with it; the same queue WITHOUT node decay "
                                   "(replay_nonlifelong_*, poses identical to the reference mapper's) is the accuracy reference"}
    # the SAME queue without node decay: the run whose poses are pinned to the reference mapper's bit for bit (replay_cpu_baseline,
    # tools/replay_vs_reference.py) and whose map is usable -- the honest end-to-end figure next to the lifelong one
    nl = replay.run(n_scans, lifelong=False, mode="sync", device=device)
    res["replay_nonlifelong_scans_per_s"] = nl["scans_per_s"]
    res["replay_nonlifelong"] = {"scans_per_s": nl["scans_per_s"], "wall_s": nl["wall_s"], "accepted": nl["accepted"], "closures": int(nl["stats"]["loop_closures"]),
                                 "map_iou_vs_truth_poses": nl["map_iou_vs_truth_poses"], "pose_error_xy_rms_m": nl["pose_error_xy_rms_m"],
                                 "graph_components": nl["graph_components"], "ms_split": nl["ms_split"]}
    if full_length:
        # config[4] at its stated size: 50 000 scans, lifelong, sync queue (13 s of replay + the queue's range noise)
        big = replay.run(50000, lifelong=True, mode="sync", device=device)
        res["replay_50k_scans_per_s"] = big["scans_per_s"]
        res["replay_50k"] = {"scans_per_s": big["scans_per_s"], "wall_s": big["wall_s"], "accepted": big["accepted"], "alive": big["alive"],
                             "closures": int(big["stats"]["loop_closures"]), "matches": int(big["stats"]["matches"]),
                             "map_iou_vs_truth_poses": big["map_iou_vs_truth_poses"], "pose_error_xy_rms_m": big["pose_error_xy_rms_m"],
                             "graph_components": big["graph_components"], "ms_split": big["ms_split"]}
    period, n_async = 0.025, 800                      # 40 Hz (UTM-30LX): 20 s of wall time when nothing is dropped
    a = replay.run(n_async, lifelong=True, mode="async", period_s=period, device=device)
    res["replay_async"] = {"period_s": period, "scans": n_async, "processed": a["processed"], "dropped": a["dropped"], "wall_s": a["wall_s"],
                           "busy_s": a["busy_s"], "waited_for_arrivals_s": a["waited_for_arrivals_s"],
                           "map_iou_vs_truth_poses": a["map_iou_vs_truth_poses"],
                           "note": "scan k arrives at k * period_s; the mapper waits for arrivals when it is ahead and skips to the most "
                                   "recent arrival when it is behind (depth-1 queue)"}
    if cpu:
        res.update(replay_cpu_baseline())
    return res


def enumeration_leg(device=0, n_scans=10000, n_queries=256):
    """Next row f-1: loop-candidate enumeration -- FindNearLinkedScans + every chain
    FindPossibleLoopClosure returns -- for a batch of query scans on the 10k-node graph (odometry chain + a
    tenth of the near-pair links, so revisited aisles hold unlinked runs).  Extra keys, rank 0."""
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.loop_search import MapperGraphSearch
    g = synth.make_pose_graph(n_scans, 3 * n_scans, seed=12345)
    xy = g["truth"][:, :2].copy()
    edges = np.concatenate([g["edges"][:n_scans - 1], g["edges"][n_scans - 1::10]])
    nbr = [[] for _ in range(n_scans)]
    for a, b in edges:
        nbr[a].append(b)
        nbr[b].append(a)
    ptr = np.zeros(n_scans + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([len(v) for v in nbr])
    idx = np.asarray([w for v in nbr for w in v], dtype=np.int32)
    s = MapperGraphSearch(device)
    s.SetGraph(xy, ptr, idx)
    queries = np.linspace(0, n_scans - 1, n_queries).astype(np.int32)
    s.FindPossibleLoopClosures(queries, 3.0, 10)
    times, kernel = [], []
    for _ in range(10):
        t = time.perf_counter()
        out = s.FindPossibleLoopClosures(queries, 3.0, 10)
        times.append(time.perf_counter() - t)
        kernel.append(s.last_kernel_ms())
    s.close()
    med = float(np.median(times))
    return {"loop_enumeration_queries_per_s": n_queries / med, "loop_enumeration_kernel_ms": float(np.median(kernel)),
            "loop_enumeration_workload": f"{n_queries} query scans x {n_scans}-scan graph ({len(edges)} edges), "
                                         f"{sum(len(c) for c in out)} chains, loop_search_maximum_distance 3.0, chain >= 10"}


def occupancy_leg(device=0, n_scans=1000):
    """Next row f-2: OccupancyGrid::CreateFromScans of `n_scans` 1081-beam scans at 5 cm (extra keys, rank 0).
    The C ABI takes host buffers, so the call time includes packing + H2D; the trace kernel time is reported
    next to it.  CPU: the C oracle (one core) on a 100-scan sample of the same queue."""
    from common import LASER
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.occupancy_grid import OccupancyGrid
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan
    world = synth.make_world(12345)
    truth, _ = synth.trajectory(2000)
    rng = np.random.default_rng(5)
    base = [synth.make_scan(world, truth[i], rng) for i in range(0, 2000, 20)]          # 100 distinct range vectors
    idx = np.linspace(0, 1999, n_scans).astype(int)
    scans = [LocalizedRangeScan(base[k % len(base)], truth[i], LASER.min_angle, LASER.ang_res) for k, i in enumerate(idx)]
    g = OccupancyGrid.CreateFromScans(scans, 0.05, LASER, device)                         # warm-up + dimensions
    times = []
    for _ in range(5):
        g.Clear()
        t0 = g.stats()["trace_ms"]
        t = time.perf_counter()
        g.AddScans(scans, LASER)
        g.Update()
        times.append((time.perf_counter() - t, g.stats()["trace_ms"] - t0))
    cells = g.cells()
    g.close()
    wall = float(np.median([a for a, _ in times]))
    kern = float(np.median([b for _, b in times]))
    out = {"occupancy_scans_per_s": n_scans / wall, "occupancy_call_ms": wall * 1e3, "occupancy_trace_kernel_ms": kern,
           "occupancy_workload": f"{n_scans} scans x {P_BEAMS} beams, {cells.shape[1]} x {cells.shape[0]} cells at 5 cm, "
                                 f"{int((cells == 100).sum())} occupied / {int((cells == 255).sum())} free"}
    try:
        from oracle import karto
        sample = [karto.Scan(s.ranges, s.GetSensorPose(), LASER) for s in scans[:100]]
        t = time.perf_counter()
        karto.occupancy_from_scans(cells.shape[1], cells.shape[0], g.offset, 0.05, sample, LASER)
        out["occupancy_cpu_scans_per_s"] = 100 / (time.perf_counter() - t)
        out["occupancy_cpu_kind"] = "port (C oracle, 1 core, 100-scan sample)"
    except Exception:
        pass
    return out


def loop_leg_devices(devices, n_pairs=256):
    """BASELINE config[2] on the devices of an N > 1 run, from ONE process: the 256 distinct loop-closure pairs dealt round robin
    over one (preset L, preset S) matcher pair per device, each pair driven by its own host thread through
    kh_loop_closure_batch (the C call releases the GIL) -- how kh_mapper_create_on_devices deals TryCloseLoop's candidate
    batches (MapperGraph::TryCloseLoop, Mapper.cpp:1500-1561).  The graph's scans are resident on the device that reads them."""
    import threading
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, LoopClosureBatch, MapperParams, ScanMatcher
    lb = synth.loop_batch(n_pairs)
    nm = len(devices)
    mp = MapperParams(**OFFLINE_PARAMS)
    members = []
    for k, dev in enumerate(devices):
        ids = list(range(k, n_pairs, nm))
        cache = {}

        def scan_at(i, dev=dev, cache=cache):
            if i not in cache:
                cache[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res).MakeResident(dev)
            return cache[i]
        queries = [LocalizedRangeScan(lb["ranges"][lb["pairs"][u][0]], lb["pairs"][u][1], LASER.min_angle, LASER.ang_res) for u in ids]
        chains = [[scan_at(i) for i in lb["pairs"][u][2]] for u in ids]
        mL = ScanMatcher.Create(mp, *PRESETS["L"]["create"], device=dev, max_batch=len(ids))
        mS = ScanMatcher.Create(mp, *PRESETS["S"]["create"], device=dev, max_batch=len(ids))
        members.append((mL, mS, ScanMatcher.pack_batch(queries, chains), cache))
    passed = [0] * nm

    def run():
        def work(k):
            mL, mS, pack, _ = members[k]
            o = LoopClosureBatch(mL, mS, None, None, LASER.min_angle, LASER.ang_res, 0.35, 9.0, pieces=1, packed=pack)
            passed[k] = int(o["passed"].sum())
        th = [threading.Thread(target=work, args=(k,)) for k in range(nm)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    run()
    times = []
    for _ in range(5):
        t = time.perf_counter()
        run()
        times.append(time.perf_counter() - t)
    for mL, mS, _, _ in members:
        mL.close(); mS.close()
    med = float(np.median(times))
    return {"loop_batch_ms": med * 1e3, "loop_pairs_per_s": n_pairs / med, "loop_batch_devices": list(map(int, devices)),
            "loop_batch_passed_gate": int(sum(passed)),
            "loop_workload": f"{n_pairs} distinct pairs dealt over {nm} (preset L, preset S) matcher pairs, one per device, from one process "
                             f"(kh_loop_closure_batch per member, a host thread each)"}


def replay_leg_devices(devices, n_scans=1500):
    """BASELINE config[4] on the devices of an N > 1 run: the lifelong replay through kh_mapper_create_on_devices (candidate
    batches of TryCloseLoop / near chains dealt over one matcher pair per device), next to the same queue on the first device
    alone: the two runs must end with the same poses bit for bit (dealing the candidates does not change which one is
    accepted first)."""
    from slam_toolbox_amd import replay
    many = replay.run(n_scans, lifelong=True, mode="sync", device=devices[0], devices=list(devices))
    one = replay.run(n_scans, lifelong=True, mode="sync", device=devices[0])
    same = bool(many["poses"].shape == one["poses"].shape and np.array_equal(many["poses"], one["poses"]))
    st = many["stats"]
    return {"replay_scans_per_s": many["scans_per_s"], "replay_scans_per_s_one_device": one["scans_per_s"],
            "replay_devices": list(map(int, devices)), "replay_poses_identical_to_one_device": same,
            "replay_workload": f"{n_scans}-scan lap circuit, lifelong mode, sync queue through kh_mapper_create_on_devices({list(devices)}): "
                               f"{many['accepted']} accepted, {many['alive']} alive, {st['loop_closures']} loop closures"}


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (one process per GPU,
    torch.distributed.run on 127.0.0.1) and hand their single JSON line through.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, KH_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# keys of the JSON line in the order they are printed: the contract's first, then every BASELINE metric, then the rest; the
# long texts (workload descriptions, notes, per-form arrays) go to the details file, not into the line
LINE_ORDER = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "solve_ms", "solve_ms_cached_analysis", "solve_ms_edge_sharded", "loop_batch_ms", "loop_pairs_per_s", "replay_scans_per_s",
              "replay_nonlifelong_scans_per_s", "replay_50k_scans_per_s",
              "match_scan_ms", "value_windows", "value_no_skipping", "value_dense_world", "roofline", "cpu_baseline", "solve_rooflines", "match_scan_roofline"]
LINE_BUDGET = 6000


def is_long_text(v):
    return isinstance(v, str) and len(v) > 60


def compact(v, depth=0):
    """the value as it goes into the line: notes, per-form arrays and long texts dropped (`sample` and `workload` cut to a
    sentence), floats rounded to 6 significant digits"""
    if isinstance(v, float):
        return float(f"{v:.6g}") if math.isfinite(v) else None
    if isinstance(v, dict):
        out = {}
        for k, x in v.items():
            if k in ("note", "forms", "traffic_source"):
                continue
            if k in ("sample", "workload") and isinstance(x, str):
                out[k] = x if len(x) <= 140 else x[:137] + "..."
            elif not is_long_text(x):
                out[k] = compact(x, depth + 1)
        return out
    if isinstance(v, (list, tuple)):
        return [compact(x, depth + 1) for x in v]
    return v


def build_line(full):
    """ONE line of at most LINE_BUDGET bytes: ordered keys first, the remaining short scalars after them while they fit."""
    line = {}
    for k in LINE_ORDER:
        if k in full and (full[k] is not None or k == "vs_baseline"):
            line[k] = compact(full[k])
    rest = [k for k in full if k not in line and not is_long_text(full[k])]
    # scalars before containers, so that a tail cut by a reader loses the least
    rest.sort(key=lambda k: (isinstance(full[k], (dict, list)), k))
    for k in rest:
        cand = dict(line)
        cand[k] = compact(full[k])
        if len(json.dumps(cand)) <= LINE_BUDGET:
            line = cand
    return line


def main():
    # ONE JSON line on stdout: everything else a library prints there (RCCL's version banner at communicator creation,
    # gloo's connection notes, karto's "Registering sensor") is pointed at stderr for the whole run
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--streams", type=int, default=1,
                    help="matcher handles (each its own HIP stream and host thread) a rank drives concurrently; "
                         "a step is still ONE batch of --batch matches on one of them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solver", action="store_true")
    ap.add_argument("--no-loop", action="store_true")
    ap.add_argument("--no-replay-50k", action="store_true", help="skip config[4] at its stated size (50 000 scans: ~25 s with the queue's making)")
    ap.add_argument("--no-variants", action="store_true", help="skip the no-skipping / dense-world variants of the headline")
    ap.add_argument("--details", default=os.path.join(ROOT, "profiles", "bench_details_latest.json"),
                    help="where the full record (every key, notes, workload texts) is written; '' = nowhere")
    ap.add_argument("--verbose", action="store_true", help="print the full record on the line instead of the compact one")
    args = ap.parse_args()

    # --gpus is honoured: N > 1 without a launcher around us -> start the N ranks; a launcher whose world size differs -> refuse
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}: refusing to report a line for the wrong size\n")
        raise SystemExit(2)

    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # last resort against a stuck collective or a teardown that never returns: after KH_BENCH_WATCHDOG seconds
    # (default 30 min, far beyond any default run) every thread's Python stack goes to stderr and the process exits
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("KH_BENCH_WATCHDOG", "1800")), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libkartohip has no CPU fallback")
    # one process per GPU.  KH_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a single-GPU box (ranks
    # then share the device); the driver's multi-GPU run uses the default, nccl = RCCL.
    backend = os.environ.get("KH_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s); RCCL needs one device per rank "
                         f"(KH_BENCH_BACKEND=gloo shares a device for functional runs)\n")
        raise SystemExit(2)
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        pg_timeout = datetime.timedelta(seconds=900)       # a collective that waits longer than this has lost a peer (rank 0 runs its one-process legs while the others wait)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)

    from common import C2_PARAMS, PRESETS, Scenario
    from slam_toolbox_amd.scan_matcher import MapperParams, ScanMatcher, _scan_array
    from slam_toolbox_amd import shard
    B = args.batch
    S = max(1, args.streams)
    corr = ((0.15, 0.15), (0.005, 0.005), math.radians(20.0), math.radians(0.5))

    def make_pairs(n_distinct, **scenario_kw):
        """(queries, centers, bases) of B pairs; n_distinct < B tiles the distinct ones"""
        qs, cs, bs = [], [], []
        for b in range(n_distinct):
            sc = Scenario(seed=1000 * rank + b, start=(37 * (rank * B + b)) % 380,
                          perturb=(0.04 * math.sin(b), -0.03 * math.cos(b), 0.01 * (b % 5 - 2)), **scenario_kw)
            q, base = sc.hip_scans()
            qs.append(q); bs.append(base); cs.append(sc.query_pose)
        idx = [b % n_distinct for b in range(B)]
        return [qs[i] for i in idx], np.asarray([cs[i] for i in idx]), [bs[i] for i in idx]

    def make_handle(queries, bases):
        h = ScanMatcher.Create(MapperParams(**C2_PARAMS), *PRESETS["C2"]["create"], device=local_rank, max_batch=B)
        for b in range(B):
            h.AddScans(queries[b], bases[b], slot=b)
        return h

    # B independent (query, chain of 10 base scans) pairs along the synthetic warehouse trajectory,
    # different per rank; grids rasterised once -> resident in HBM.  Every handle holds the same B pairs.
    queries, centers, bases = make_pairs(B, n_base=10)
    handles = [make_handle(queries, bases) for _ in range(S)]
    arr = (_scan_array(queries), B)

    def step(h, c=centers, a=arr):
        return h.CorrelateScanBatch(None, c, *corr, True, False, scan_array=a)

    def timed(hs, n_steps, c=centers, a=arr):
        """n_steps steps dealt round-robin to the handles `hs` (one host thread each); returns (seconds, per-step seconds of the
        first handle's thread, last result).  Barrier + synchronize on both sides; max over ranks."""
        import threading
        results = [None] * len(hs)
        per_step = []

        def worker(k):
            out = None
            for _ in range(k, n_steps, len(hs)):
                t = time.perf_counter()
                out = step(hs[k], c, a)
                if k == 0:
                    per_step.append(time.perf_counter() - t)
            results[k] = out
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(hs))]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        for out in results:
            if out is not None:
                resp, means, covs, status = out
                # (KH_BENCH_NO_CHECK: measurement builds whose kernels skip work on purpose, tools/build_variant.sh)
                assert os.environ.get("KH_BENCH_NO_CHECK") or ((status == 0).all() and (resp > 0.1).all()), "matches failed"
        return (shard.max_over_ranks(dt, device="cuda") if world > 1 else dt), per_step

    # (untimed setup in front of the W warm-up steps: half a second of steps, so that the device's clocks and the host pool's threads
    # are where a running service has them -- on some boxes of the pool the first 50 steps behind a 5-step warm-up ran at 88 k
    # matches/s and every window behind them at 103 k; KH_BENCH_PREROLL_S=0 switches it off)
    preroll_s = float(os.environ.get("KH_BENCH_PREROLL_S", "0.5"))
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < preroll_s:
        for h in handles:
            step(h)
    for _ in range(args.warmup):
        for h in handles:
            step(h)
    # THE timed region: exactly K steps, nothing but the work in it.  (Through round 4 the library's profiling events were on
    # during it: two records per chunk around the scoring kernel cost 10 % of the step -- 74 k with them, 87 k without, same box.)
    dt, per_step = timed(handles, args.steps)
    # `value` is the K steps asked for; beside it a region of at least 160 steps (half a second) in five windows, so that the
    # headline can be read against the box-to-box spread (a 20-step region is 0.07 s)
    long_steps = max(args.steps, 160)
    dt_long, per_step_long = (dt, per_step) if long_steps == args.steps else timed(handles, long_steps)
    # the roofline's kernel durations: the same K steps once more with HIP events on the library's streams around the table,
    # scoring and tie kernels of every launch (kh_matcher_profile / kh_matcher_profile_side)
    for h in handles:
        h.profile(True)
    dt_profiled, _ = timed(handles, args.steps)
    wave_loads = sum(h.score_loads() for h in handles)
    sides = [h.profile_side() for h in handles]
    profs = [h.profile(False) for h in handles]
    prof = {k: sum(p[k] for p in profs) for k in profs[0]}
    side = {k: sum(p[k] for p in sides) for k in sides[0]}

    full = {}          # every key of the record; the line is cut from it
    variants = {}
    if not args.no_variants and S == 1:
        # (a) the same batch with empty-window skipping switched off (kh_matcher_set_debug bit 2): every one of the 81 x 1081 windows
        # of a match is scored whatever the grid holds -- the strict worst case, what a world without free space would cost
        n_var = max(5, args.steps // 4)
        hm = handles[0]
        hm.set_debug(False, dense_score=True)
        step(hm)
        hm.profile(True)
        dtv, _ = timed([hm], n_var)
        loads_v = hm.score_loads()
        pv = hm.profile(False)
        hm.set_debug(False)
        variants["value_no_skipping"] = world * B * n_var / dtv
        variants["no_skipping"] = {"steps": n_var, "ms_per_step": dtv / n_var * 1e3, "k3_launch_ms": pv["score_ms"] / max(1, pv["score_launches"]),
                                   "window_reads_per_launch": loads_v / max(1, pv["score_launches"])}
        # (b) a denser world: 400 pillars, 40 base scans five poses apart -> 1.5 % of the grid non-zero (default 0.45 %); 16 distinct
        # pairs tiled over the batch (the slots are what the kernel sees: 256 grids either way)
        try:
            q2, c2, b2 = make_pairs(16, n_base=40, step=5, n_pillars=400, n_traj=600)
            h2 = make_handle(q2, b2)
            a2 = (_scan_array(q2), B)
            for _ in range(2):
                step(h2, c2, a2)
            h2.profile(True)
            dtd, _ = timed([h2], n_var, c2, a2)
            loads_d = h2.score_loads()
            pd = h2.profile(False)
            nz = None
            try:
                g = h2.GetCorrelationGrid(0)
                nz = float((np.asarray(g) != 0).mean())
            except Exception:
                pass
            h2.close()
            variants["value_dense_world"] = world * B * n_var / dtd
            variants["dense_world"] = {"steps": n_var, "ms_per_step": dtd / n_var * 1e3, "k3_launch_ms": pd["score_ms"] / max(1, pd["score_launches"]),
                                       "window_reads_per_launch": loads_d / max(1, pd["score_launches"]), "grid_nonzero_frac": nz,
                                       "distinct_pairs": 16, "base_scans": 40, "pillars": 400}
        except Exception as exc:
            variants["dense_world_error"] = repr(exc)[:200]

    solver_out = None
    if not args.no_solver:
        # the sharded solve is a collective: every rank takes part, rank 0 reports
        if world > 1:
            dist.barrier()
        if world > 1 or rank == 0:
            try:
                solver_out = solver_leg(local_rank, rank, world, cpu=not args.no_cpu_baseline)
            except Exception as exc:      # the headline line must survive a failure of the extra leg
                solver_out = {"solver_leg_error": repr(exc)[:200]}
    strong_out = None
    if not args.no_loop:
        if world > 1:
            dist.barrier()
        try:
            strong_out = strong_leg(local_rank, rank, world)
        except Exception as exc:
            strong_out = {"strong_leg_error": repr(exc)[:200]}
        # the in-process form (what the mapper front end uses): rank 0 drives one member per GPU of the run from ONE
        # process, the other ranks wait; at N = 1 also two members sharing the GPU (the N > 1 code path on one device)
        if world > 1:
            dist.barrier()
        if rank == 0:
            devices = list(range(world)) if backend == "nccl" else [local_rank] * world
            try:
                strong_out["strong_scaling_in_process"] = group_leg(devices)
                if world == 1:
                    strong_out["strong_scaling_in_process_two_members_one_gpu"] = group_leg([0, 0])
            except Exception as exc:
                strong_out["group_leg_error"] = repr(exc)[:200]
            if world > 1:
                # config[2] and config[4] on the devices of the run, from this one process: the loop-closure batch dealt over one
                # (preset L, preset S) matcher pair per device, and the lifelong replay through kh_mapper_create_on_devices
                try:
                    strong_out.update(loop_leg_devices(devices))
                except Exception as exc:
                    strong_out["loop_leg_devices_error"] = repr(exc)[:200]
                try:
                    strong_out.update(replay_leg_devices(devices))
                except Exception as exc:
                    strong_out["replay_leg_devices_error"] = repr(exc)[:200]
        if world > 1:
            dist.barrier()
    if rank == 0:
        k3_ms = prof["score_ms"] / max(1, prof["score_launches"])
        # a step's batch is scored in sub-batches (pipelined with the host half): matches per scoring launch
        per_launch = B * args.steps / max(1, prof["score_launches"])
        alg = ALG_BYTES_C2 * per_launch
        alg_gbs = alg / (k3_ms * 1e-3) / 1e9
        # LDS side, measured live: K2' tallies on the device the wave-level ds_read_b32 (256 B each) K3' issues for the windows it keeps
        loads_per_launch = wave_loads / max(1, prof["score_launches"])
        lds_gbs = loads_per_launch * 256.0 / (k3_ms * 1e-3) / 1e9
        traffic = pmc_traffic(per_launch)
        chunk = max(1, len(per_step_long) // 5)
        windows = [per_step_long[i * chunk:(i + 1) * chunk] for i in range(5)] if len(per_step_long) >= 5 else [per_step_long]
        rates = sorted(world * B * len(w) / sum(w) for w in windows if w)
        # reference lookups per CU clock (one figure that does not change its denominator from round to round):
        # every lookup of the reference's access stream (nX * nY * nA * P per match) over the scoring kernel's time, and the part
        # of them the kernel really reads (windows it keeps: 64 x 64 bytes each, of which 61 x 61 are poses)
        cu_clocks = 256 * 2.4e9 * (k3_ms * 1e-3)
        lookups_all = C2["nx"] * C2["ny"] * C2["na"] * P_BEAMS * per_launch
        lookups_read = loads_per_launch / 16.0 * C2["nx"] * C2["ny"]
        full.update({
            "metric": "scan-matches/sec", "value": world * B * args.steps / dt, "unit": "scan-matches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 gather / i32 sum / f64 penalty",
            "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: CorrelateScan 1081 beams 61x61x81 poses", "matches_per_step_per_gpu": B,
                       "parallelism": f"{world} x independent match shards (no collective)", "streams_per_gpu": S,
                       "untimed_preroll_s": preroll_s},
            "config_workload": "BASELINE config[1]: single-scan CorrelateScan, 1081 beams, 0.3m x 0.3m x +-20deg @ 5mm/0.5deg "
                               "(61x61x81 poses), 8087^2 grid",
            # per-step wall times of the timed region, dealt into five interleaved windows: spread of the headline
            "value_windows": {"n": len(rates), "median": float(np.median(rates)) if rates else None, "min": rates[0] if rates else None,
                              "max": rates[-1] if rates else None, "steps_per_window": len(windows[0]) if windows else 0,
                              "steps": long_steps, "seconds": dt_long, "value": world * B * long_steps / dt_long},
            # the dominant kernel reads LDS-resident window unions: the resource it leans on is the LDS read port (ds_read_b32:
            # 128 B / clk / CU), not HBM.  `frac` is the live LDS->register byte rate against that port; the HBM side (recorded
            # PMC) and the reference's algorithmic access stream are kept beside it, each labelled.
            "roofline": {"bound": "lds", "kernel": SCORE_KERNEL, "achieved": lds_gbs, "peak": LDS_B32_PEAK_GBS,
                         "unit": "GB/s", "frac": lds_gbs / LDS_B32_PEAK_GBS,
                         "window_reads_per_launch": loads_per_launch, "avg_launch_ms": k3_ms, "matches_per_launch": per_launch,
                         "ms_per_step_with_events": dt_profiled / args.steps * 1e3,
                         "lookups_per_cu_clk": lookups_all / cu_clocks, "lookups_per_cu_clk_read": lookups_read / cu_clocks,
                         "lookups_per_cu_clk_peak_b32": 128.0 * C2["nx"] * C2["ny"] / 4096.0, "lds_array_frac": lds_gbs / (2.0 * LDS_B32_PEAK_GBS),
                         "side_kernels_ms_per_launch": {"k_offsets_lds": side["offsets_ms"] / max(1, prof["score_launches"]),
                                                        "k_ties": side["ties_ms"] / max(1, prof["score_launches"])},
                         "traffic": traffic, "hbm_frac": (traffic / (k3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "algorithmic_bytes_per_launch": alg, "algorithmic_gbs": alg_gbs, "algorithmic_ratio": alg_gbs / HBM_PEAK_GBS,
                         "traffic_source": "recorded: " + os.path.relpath(PMC_FILE, ROOT) + " (rocprofv3 PMC passes of this command; "
                                           "(2*FETCH_SIZE + WRITE_SIZE)*1024, scaled to this run's matches per launch)",
                         "note": "bound = the LDS read port.  A window (one beam at one angle: 64 bytes x 64 rows) is read from a staged LDS region "
                                 "with sixteen wave-level ds_read_b32 of 256 B; `achieved` = windows kept x 4 KB over the launch time, `peak` = 128 B/clk/CU "
                                 "x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS table); the byte sums run on the matrix cores (v_mfma_i32_16x16x64_i8, "
                                 "one per KB read).  algorithmic_ratio = the reference's own access stream (5 B per lookup, every lookup) over the "
                                 "launch time against 8 TB/s: it exceeds 1 because the kernel reads LDS-resident windows, 4 lookups per dword, and "
                                 "skips windows that hold only zeros -- it is not a fraction of a hardware limit"},
        })
        full.update(variants)
        if world == 1 and not args.no_cpu_baseline:
            with _StdoutToStderr():
                full["cpu_baseline"] = cpu_baseline()
        if solver_out:
            full.update(solver_out)
        if strong_out:
            full.update(strong_out)
        if world == 1 and not args.no_loop:
            full.update(loop_leg(local_rank, cpu=not args.no_cpu_baseline))
            full.update(enumeration_leg(local_rank))
            full.update(occupancy_leg(local_rank))
            try:
                full.update(replay_leg(local_rank, cpu=not args.no_cpu_baseline, full_length=not args.no_replay_50k))
                full.update(latency_leg(local_rank))
            except Exception as exc:
                full["replay_leg_error"] = repr(exc)[:200]
        if args.details:
            try:
                with open(args.details, "w") as f:
                    json.dump(full, f, indent=1, sort_keys=True, default=str)
                full["details_file"] = os.path.relpath(args.details, ROOT)
            except OSError:
                pass
        line = full if args.verbose else build_line(full)
        json_out.write(json.dumps(line, default=str) + "\n")
        json_out.flush()
    for h in handles:
        h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    sys.stderr.flush()
    if world > 1:
        # the line is out and every handle is closed: leave without the interpreter / HIP runtime / thread-pool
        # teardown (an N = 2 run was once seen to sit in it until the launcher's timeout).  Single-process runs exit
        # normally so that a profiler wrapped around them (rocprofv3) can write its output.
        os._exit(0)


if __name__ == "__main__":
    main()
