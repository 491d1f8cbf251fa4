"""GPU parity of the HIP SPA solver (through the C ABI) against the CPU restatement oracle/spa.py.
Tolerance: node poses within 1e-7 of the oracle in the Ceres-like configuration (same LM trajectory, same
iteration count; the bar is set by conditioning, not by the algorithm: the CPU oracle run twice with two
different SuperLU column orderings differs from itself by 5e-10 on the 2000-node graph, and every LM
iteration re-amplifies the rounding of a linear solve with cond ~ 1e7); in the `tight` configuration the optimum's cost to 1e-11 relative and the poses to 1e-5
(north_star asks for 1e-4 m / 1e-4 rad against the reference; the oracle itself is "parity unpinned" at
the Ceres boundary, see oracle/spa.py)."""
import numpy as np
import pytest

from slam_toolbox_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-7      # m / rad; north_star's bar against the reference is 1e-4
TIGHT = dict(max_num_iterations=200, function_tolerance=1e-15, gradient_tolerance=1e-14, parameter_tolerance=1e-14)


def _diff(a, b):
    d = np.asarray(a) - np.asarray(b)
    d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return float(np.abs(d).max())


@pytest.mark.parametrize("n,e,seed", [(50, 80, 1), (300, 700, 5), (2000, 5000, 9)])
def test_solver_matches_oracle(kartohip_lib, n, e, seed):
    from oracle import spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(n, e, seed=seed)
    sol = HipSpaSolver()
    # Ceres-like options (ceres_solver.cpp:157-186)
    ref_x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    assert summ["usable"] == 1
    assert summ["iterations"] == info["iterations"], (summ, info["iterations"], info["message"])
    assert _diff(sol.poses(), ref_x) < POSE_TOL
    assert abs(summ["final_cost"] - info["final_cost"]) <= 1e-9 * max(1.0, info["final_cost"])
    # corrections = all nodes (ceres_solver.cpp:256-268)
    corr = sol.GetCorrections()
    assert len(corr) == n and _diff(np.asarray([p for _, p in corr]), ref_x) < POSE_TOL
    # tight
    ref_t, info_t = spa.solve(g["init"], g["edges"], g["z"], g["cov"], spa.Options.tight())
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    sol.Configure(TIGHT)
    summ = sol.Compute()
    # `tight` runs end where the cost stops changing in FP64; the weakest modes of a pose graph leave the
    # poses undetermined at the ~1e-6 level inside that plateau (two CPU runs with function_tolerance
    # 1e-15 / 0 differ by 1.4e-6 on this graph), so the pinned quantities are the optimum's cost, the
    # gradient at the returned poses, and the poses well inside north_star's 1e-4 bar.
    assert abs(summ["final_cost"] - info_t["final_cost"]) <= 1e-11 * info_t["final_cost"]
    prob = spa.Problem(sol.poses(), g["edges"], g["z"], g["cov"])
    _, grad, _ = prob.linearize(prob.x)
    assert np.abs(grad).max() < 1e-6
    assert _diff(sol.poses(), ref_t) < 1e-5
    sol.close()


def _clique_graph(n_clique, n_chain, seed):
    """n_clique mutually linked poses on a circle (one dense front of 3*n_clique - 3 pivots: every panel
    width, partial last panel, partial MFMA tiles) followed by an odometry chain."""
    import math
    rng = np.random.default_rng(seed)
    n = n_clique + n_chain
    truth = np.zeros((n, 3))
    for i in range(n_clique):
        a = 2 * math.pi * i / n_clique
        truth[i] = [3 * math.cos(a), 3 * math.sin(a), a + 1.0]
    for i in range(n_clique, n):
        truth[i] = truth[i - 1] + [0.4 * math.cos(0.1 * i), 0.4 * math.sin(0.1 * i), 0.05]
    pairs = [(i, j) for i in range(n_clique) for j in range(i + 1, n_clique)]
    pairs += [(i - 1, i) for i in range(n_clique, n)]
    z = np.zeros((len(pairs), 3))
    for e, (a, b) in enumerate(pairs):
        c, s_ = math.cos(truth[a, 2]), math.sin(truth[a, 2])
        dx, dy = truth[b, 0] - truth[a, 0], truth[b, 1] - truth[a, 1]
        z[e] = [c * dx + s_ * dy + rng.normal(0, 0.01), -s_ * dx + c * dy + rng.normal(0, 0.01),
                (truth[b, 2] - truth[a, 2] + rng.normal(0, 0.003) + math.pi) % (2 * math.pi) - math.pi]
    cov = np.tile(np.diag([1e-3, 2e-3, 4e-4]).reshape(9), (len(pairs), 1))
    init = truth + rng.normal(0, 0.05, truth.shape)
    init[0] = truth[0]
    return dict(init=init, edges=np.asarray(pairs, dtype=np.int32), z=z, cov=cov)


@pytest.mark.parametrize("n_clique,n_chain", [(6, 0), (23, 5), (70, 40), (150, 3)])
def test_dense_fronts(kartohip_lib, n_clique, n_chain):
    from oracle import spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = _clique_graph(n_clique, n_chain, seed=n_clique)
    ref_x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
    sol = HipSpaSolver()
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    assert summ["usable"] == 1 and summ["iterations"] == info["iterations"], (summ, info["iterations"])
    assert _diff(sol.poses(), ref_x) < POSE_TOL
    sol.close()


def test_noise_free_graph_returns_ground_truth(kartohip_lib):
    from slam_toolbox_amd.scan_solver import HipSpaSolver, link_info
    g = synth.make_pose_graph(400, 900, seed=3)
    z = np.asarray([link_info(g["truth"][a], g["truth"][b], np.eye(3))[0] for a, b in g["edges"]])
    sol = HipSpaSolver(options=TIGHT)
    sol.load(g["init"], g["edges"], z, g["cov"])
    sol.Compute()
    # gauge: node 0 stays where it was (= truth[0])
    assert _diff(sol.poses(), g["truth"]) < 1e-9
    sol.close()


def test_plugin_api_semantics(kartohip_lib):
    """ModifyNode adds the old yaw (ceres_solver.cpp:457-459); unknown ids are reported, not fatal;
    RemoveConstraint tries (a, b) then (b, a); RemoveNode drops the node's constraints."""
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    s = HipSpaSolver()
    for i in range(4):
        s.AddNode(10 + i, [float(i), 0.0, 0.1 * i])
    s._ids = [10, 11, 12, 13]
    cov = np.diag([1e-3, 1e-3, 4e-4])
    for i in range(3):
        s.AddConstraint(10 + i, 11 + i, [1.0, 0.0, 0.1], cov)
    s.AddConstraint(10, 99, [1.0, 0.0, 0.0], cov)
    assert "could not find nodes" in s.last_warning
    s.ModifyNode(12, [5.0, 6.0, 0.3])
    assert abs(s.GetNodeOrientation(12) - 0.5) < 1e-15
    assert capi_count(s) == (4, 3)
    s.RemoveConstraint(12, 11)          # stored as (11, 12): found through the swapped lookup
    assert capi_count(s) == (4, 2)
    s.RemoveNode(13)
    assert capi_count(s) == (3, 1)
    s.RemoveNode(77)
    assert "Failed to find node" in s.last_warning
    summ = s.Compute()
    assert summ["usable"] == 1
    s.Clear()
    assert s.GetCorrections() == []
    s.Reset()
    assert capi_count(s) == (0, 0)
    s.Compute()
    assert "no nodes" in s.last_warning
    s.close()


def capi_count(s):
    from slam_toolbox_amd import capi
    return capi.lib().kh_spa_num_nodes(s._h), capi.lib().kh_spa_num_constraints(s._h)


def test_link_info_golden(kartohip_lib):
    """kh_link_info against the reference's LinkInfo::Update (tests/golden/link_info.npz)."""
    import os
    from slam_toolbox_amd.scan_solver import link_info
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "link_info.npz"))
    for i in range(g["pose1"].shape[0]):
        d, c = link_info(g["pose1"][i], g["pose2"][i], g["cov"][i])
        assert np.array_equal(d, g["diff"][i])
        assert np.array_equal(c, g["cov_out"][i])


def test_disconnected_components_and_topology_edits(kartohip_lib):
    """Two components (only the first holds the gauge node: the second is held by the LM damping alone, like in
    Ceres) and a solve after RemoveConstraint / RemoveNode changed the topology: same poses as the oracle."""
    from oracle import spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g1 = synth.make_pose_graph(120, 260, seed=51)
    g2 = synth.make_pose_graph(90, 190, seed=52)
    init = np.concatenate([g1["init"], g2["init"] + [100.0, 0.0, 0.0]])
    edges = np.concatenate([g1["edges"], g2["edges"] + 120])
    z = np.concatenate([g1["z"], g2["z"]])
    cov = np.concatenate([g1["cov"], g2["cov"]])
    sol = HipSpaSolver()
    sol.load(init, edges, z, cov)
    summ = sol.Compute()
    ref_x, info = spa.solve(init, edges, z, cov)
    assert summ["usable"] == 1 and summ["iterations"] == info["iterations"]
    assert _diff(sol.poses(), ref_x) < POSE_TOL
    # drop the last 40 constraints and the last node, solve again from the current state
    cur = sol.poses()
    for a, b in edges[-40:]:
        sol.RemoveConstraint(int(a), int(b))
    sol.RemoveNode(209)
    sol._ids = list(range(209))
    keep = [k for k in range(len(edges) - 40) if 209 not in (edges[k, 0], edges[k, 1])]
    summ = sol.Compute()
    ref_x, info = spa.solve(cur[:209], edges[keep], z[keep], cov[keep])
    assert summ["usable"] == 1 and summ["iterations"] == info["iterations"]
    assert _diff(sol.poses(), ref_x) < POSE_TOL
    sol.close()


@pytest.mark.parametrize("loss", ["HuberLoss", "CauchyLoss"])
def test_robust_loss_matches_oracle(kartohip_lib, loss):
    """`ceres_loss_function` (ceres_solver.cpp:82-94) with false loop closures in the graph: same LM
    trajectory as the oracle's corrected linearisation."""
    from oracle import spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(300, 700, seed=41)
    rng = np.random.default_rng(42)
    z = g["z"].copy()
    loops = np.flatnonzero(np.abs(g["edges"][:, 0] - g["edges"][:, 1]) > 1)
    bad = rng.choice(loops, size=20, replace=False)
    z[bad, :2] += rng.normal(0, 1.5, (20, 2))
    z[bad, 2] += rng.normal(0, 0.4, 20)
    opt = spa.Options(); opt.loss_function = loss
    ref_x, info = spa.solve(g["init"], g["edges"], z, g["cov"], opt)
    sol = HipSpaSolver(options=dict(loss_function=loss))
    sol.load(g["init"], g["edges"], z, g["cov"])
    summ = sol.Compute()
    assert summ["usable"] == 1 and summ["iterations"] == info["iterations"], (summ, info["iterations"])
    assert abs(summ["initial_cost"] - info["initial_cost"]) <= 1e-12 * info["initial_cost"]
    assert abs(summ["final_cost"] - info["final_cost"]) <= 1e-9 * info["final_cost"]
    assert _diff(sol.poses(), ref_x) < POSE_TOL
    # and it is not the squared-loss answer
    ref_sq, _ = spa.solve(g["init"], g["edges"], z, g["cov"])
    assert _diff(ref_sq, ref_x) > 1e-3
    sol.close()


def _solve(g, **kw):
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    sol = HipSpaSolver()
    sol.set_debug(**kw)
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    x = sol.poses()
    sol.close()
    return summ, x


@pytest.mark.parametrize("n_clique,n_chain", [(2, 0), (4, 3), (6, 9), (11, 0), (12, 2), (17, 30), (22, 1), (23, 5), (33, 7), (43, 2), (44, 40), (70, 40),
                                              (86, 1), (127, 3), (150, 3)])
def test_linear_solves_of_dense_fronts(kartohip_lib, n_clique, n_chain):
    """Every tail shape of the pivot block (3 * (n_clique - 1) pivots: 3 ... 447; above 126 the supernode is a chain of
    fronts) through the level pipeline: the residual of EVERY linear solve of the run, evaluated from the block-sparse
    matrix, and the same poses as the panel-pair kernels."""
    g = _clique_graph(n_clique, n_chain, seed=100 + n_clique)
    s3, x3 = _solve(g, check_linear_solves=True, factor_kernels=3)
    s2, x2 = _solve(g, check_linear_solves=True, factor_kernels=2)
    assert s3["usable"] == 1 and s3["iterations"] == s2["iterations"], (s3, s2)
    assert 0.0 < s3["worst_linear_residual"] < 1e-9, s3
    assert 0.0 < s2["worst_linear_residual"] < 1e-9, s2
    assert _diff(x3, x2) < 1e-9


@pytest.mark.parametrize("n,e,seed", [(40, 60, 3), (300, 700, 5), (2000, 5000, 9), (3000, 9000, 4)])
def test_linear_solves_of_pose_graphs(kartohip_lib, n, e, seed):
    g = synth.make_pose_graph(n, e, seed=seed)
    s3, x3 = _solve(g, check_linear_solves=True, factor_kernels=3)
    s2, x2 = _solve(g, check_linear_solves=True, factor_kernels=2)
    assert s3["usable"] == 1 and s3["iterations"] == s2["iterations"], (s3, s2)
    assert 0.0 < s3["worst_linear_residual"] < 1e-9, s3
    assert _diff(x3, x2) < 1e-8
    # the level pipeline with the children's update matrices read in place (no extend-add launches)
    sg, xg = _solve(g, check_linear_solves=True, factor_kernels=3, gather_children=True)
    assert sg["iterations"] == s3["iterations"] and 0.0 < sg["worst_linear_residual"] < 1e-9, sg
    assert _diff(xg, x3) < 1e-8
    # ... and with an extend-add launch per level (rounds 3-5); the default adds a front's update matrix into its parent in k_syrk
    se, xe = _solve(g, check_linear_solves=True, factor_kernels=3, extend_add_pass=True)
    assert se["iterations"] == s3["iterations"] and 0.0 < se["worst_linear_residual"] < 1e-9, se
    assert _diff(xe, x3) < 1e-8
    # the default path is bit-reproducible from run to run (two children of one level add into DIFFERENT buffers of the parent)
    s3b, x3b = _solve(g, check_linear_solves=True, factor_kernels=3)
    assert np.array_equal(x3, x3b) and s3b["final_cost"] == s3["final_cost"]


def test_incremental_reanalysis_after_a_loop_closure(kartohip_lib, monkeypatch):
    """A solved graph grows by a stretch of new scans and a few loop-closure links, and loses a node: the next Compute()
    reuses the supernodes of the last nested dissection (new nodes as leading leaves, summary.analysis == 2) and must return
    what a solver that sees the final graph for the first time returns -- same iterations, same poses to rounding.
    (The level guard of round 6 -- a re-analysis whose tree is more than two levels taller than the last dissection's goes back
    to a full dissection -- would send this one back: it is switched off here, and has its own test below.)"""
    monkeypatch.setenv("KH_SPA_EXTRA_LEVELS", "100")
    from oracle import spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(1500, 4000, seed=21)
    n0, extra = 1470, 30                                   # the last 30 nodes arrive after the first solve
    e = g["edges"]
    first = (e[:, 0] < n0) & (e[:, 1] < n0)
    a = HipSpaSolver()
    a.set_debug(check_linear_solves=True)
    for i in range(n0):
        a.AddNode(i, g["init"][i])
    for k in np.flatnonzero(first):
        a.AddConstraint(int(e[k, 0]), int(e[k, 1]), g["z"][k], g["cov"][k].reshape(3, 3))
    s1 = a.Compute()
    assert s1["analysis"] == 1 and s1["usable"] == 1
    x1 = np.array([p for _, p in a.GetCorrections()])
    for i in range(n0, n0 + extra):
        a.AddNode(i, g["init"][i])
    for k in np.flatnonzero(~first):
        a.AddConstraint(int(e[k, 0]), int(e[k, 1]), g["z"][k], g["cov"][k].reshape(3, 3))
    a.RemoveNode(700)
    s2 = a.Compute()
    assert s2["analysis"] == 2, s2
    assert 0.0 < s2["worst_linear_residual"] < 1e-9
    # the same final graph, same starting point, analysed from scratch
    b = HipSpaSolver()
    start = np.vstack([x1, g["init"][n0:n0 + extra]])
    keep = [i for i in range(n0 + extra) if i != 700]
    for i in keep:
        b.AddNode(i, start[i])
    for k in range(e.shape[0]):
        if 700 not in (int(e[k, 0]), int(e[k, 1])):
            b.AddConstraint(int(e[k, 0]), int(e[k, 1]), g["z"][k], g["cov"][k].reshape(3, 3))
    s3 = b.Compute()
    assert s3["analysis"] == 1 and s3["iterations"] == s2["iterations"]
    ids_a = [i for i, _ in a.GetCorrections()]
    assert ids_a == keep
    pa = np.array([p for _, p in a.GetCorrections()])
    pb = np.array([p for _, p in b.GetCorrections()])
    assert _diff(pa, pb) < 1e-8
    # unchanged topology: no analysis at all
    s4 = a.Compute()
    assert s4["analysis"] == 0
    a.close(); b.close()


def test_level_guard_sends_a_tall_reanalysis_back_to_a_full_dissection(kartohip_lib, monkeypatch):
    """The same growth with the guard at its tightest (no extra level allowed): the re-analysis is discarded, the solver dissects
    from scratch (summary.analysis == 1) and returns the same poses as with the incremental tree -- the guard only picks between two
    valid elimination orders."""
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(1500, 4000, seed=21)
    n0, extra = 1470, 30
    e = g["edges"]
    first = (e[:, 0] < n0) & (e[:, 1] < n0)
    poses = {}
    for guard in ("100", "0"):
        monkeypatch.setenv("KH_SPA_EXTRA_LEVELS", guard)
        a = HipSpaSolver()
        for i in range(n0):
            a.AddNode(i, g["init"][i])
        for k in np.flatnonzero(first):
            a.AddConstraint(int(e[k, 0]), int(e[k, 1]), g["z"][k], g["cov"][k].reshape(3, 3))
        assert a.Compute()["analysis"] == 1
        for i in range(n0, n0 + extra):
            a.AddNode(i, g["init"][i])
        for k in np.flatnonzero(~first):
            a.AddConstraint(int(e[k, 0]), int(e[k, 1]), g["z"][k], g["cov"][k].reshape(3, 3))
        s2 = a.Compute()
        assert s2["analysis"] == (2 if guard == "100" else 1), s2
        assert s2["usable"] == 1
        poses[guard] = np.array([p for _, p in a.GetCorrections()])
        a.close()
    assert _diff(poses["100"], poses["0"]) < 1e-8


@pytest.mark.gpu
def test_iteration_log_describes_the_compute(kartohip_lib):
    """kh_spa_iteration_log: one row per trust-region iteration of the last Compute(), consistent with its summary (the trace that
    is laid beside Ceres' IterationSummary in tests/test_spa_vs_ceres.py where Ceres exists)."""
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    g = synth.make_pose_graph(400, 1000, seed=3)
    sol = HipSpaSolver()
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    log = sol.iteration_log()
    sol.close()
    assert len(log) == summ["iterations"] and np.array_equal(log[:, 0], np.arange(1, len(log) + 1))
    assert int((log[:, 7] == 1.0).sum()) == summ["successful_steps"] - 1          # the start counts as a successful step
    acc = log[log[:, 7] == 1.0]
    assert (acc[:, 3] > 0).all() and (log[:, 4] > 0).all()                       # model decrease and radius positive
    assert abs(log[0, 1] - summ["initial_cost"]) <= 1e-12 * summ["initial_cost"]


def test_a_larger_graph_over_the_same_buffers_reads_nothing_stale(kartohip_lib):
    """The self-cleaning fronts are zeroed once per ALLOCATION: a graph whose fronts are larger than the last one's but still fit the
    buffers (they grow by half when they grow) must find zeros behind the old extent too.  Run with poisoned allocations
    (KH_SPA_POISON: new device buffers start as NaN), a solve of the larger graph after a smaller one must equal a fresh solver's
    bit for bit -- round 6's first form zeroed the old extent only, and a mapper's 7th Compute came out 2e-3 off."""
    import subprocess, sys, os
    code = """
import numpy as np
from slam_toolbox_amd import synth
from slam_toolbox_amd.scan_solver import HipSpaSolver
def solve(sol, g):
    sol.load(g["init"], g["edges"], g["z"], g["cov"]); s = sol.Compute(); return s, sol.poses()
graphs = [synth.make_pose_graph(n, e, seed=21) for n, e in ((400, 900), (440, 1000), (470, 1080), (520, 1200))]
a = HipSpaSolver(); a.set_debug(check_linear_solves=True)
for g in graphs:
    sa, xa = solve(a, g)
    b = HipSpaSolver(); sb, xb = solve(b, g); b.close()
    assert sa["usable"] == 1 and sa["iterations"] == sb["iterations"], (sa, sb)
    assert np.array_equal(xa.view(np.uint64), xb.view(np.uint64)), np.abs(xa - xb).max()
print("ok")
"""
    env = dict(os.environ, KH_SPA_POISON="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
