"""CPU: independent checks of the SPA oracle (oracle/spa.py).  The Ceres boundary is "parity unpinned"
(Ceres is not in the reference tree); what can be pinned is pinned here: LinkInfo::Update and
Matrix3::Inverse against known answers from the reference build, the optimum against scipy's generic
least-squares, and noise-free graphs against their ground truth."""
import os

import numpy as np
import pytest

from oracle import spa
from slam_toolbox_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "link_info.npz")


def _diff(a, b):
    d = np.asarray(a) - np.asarray(b)
    d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return float(np.abs(d).max())


def test_link_info_and_inverse_against_reference_known_answers():
    g = np.load(GOLD)
    for i in range(g["pose1"].shape[0]):
        d, c = spa.link_info(g["pose1"][i], g["pose2"][i], g["cov"][i])
        assert np.array_equal(d, g["diff"][i])
        assert np.array_equal(c, g["cov_out"][i])
        assert np.array_equal(spa.matrix3_inverse(g["cov_out"][i]), g["inverse"][i])


def test_tight_solution_is_the_least_squares_optimum():
    from scipy.optimize import least_squares
    g = synth.make_pose_graph(120, 260, seed=21)
    xt, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"], spa.Options.tight())
    prob = spa.Problem(g["init"], g["edges"], g["z"], g["cov"])

    def fun(v):
        xx = g["init"].copy()
        xx[prob.free_nodes] = v.reshape(-1, 3)
        r, _ = spa._residuals(xx, prob.edges[:, 0], prob.edges[:, 1], prob.z, prob.U)
        return r.reshape(-1)
    res = least_squares(fun, g["init"][prob.free_nodes].reshape(-1), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    xs = g["init"].copy()
    xs[prob.free_nodes] = res.x.reshape(-1, 3)
    assert _diff(xs, xt) < 1e-6          # scipy's trf stops at ~1e-7; the costs agree to 1e-9 below
    assert abs(res.cost - info["final_cost"]) < 1e-9 * info["final_cost"]


def test_noise_free_graph_returns_ground_truth():
    g = synth.make_pose_graph(200, 450, seed=22)
    z = np.asarray([spa.link_info(g["truth"][a], g["truth"][b], np.eye(3))[0] for a, b in g["edges"]])
    x, info = spa.solve(g["init"], g["edges"], z, g["cov"], spa.Options.tight())
    assert _diff(x, g["truth"]) < 1e-9


def test_ceres_like_options_stop_early_and_gauge_is_fixed():
    g = synth.make_pose_graph(300, 700, seed=5)
    x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
    assert info["termination"] == "CONVERGENCE" and info["iterations"] < 50
    assert np.array_equal(x[0], g["init"][0])                 # first node constant (ceres_solver.cpp:228-241)
    assert info["final_cost"] < info["initial_cost"]


def _with_outliers(g, n_bad, seed):
    """Corrupt n_bad non-odometry constraints (false loop closures)."""
    rng = np.random.default_rng(seed)
    z = g["z"].copy()
    loops = np.flatnonzero(np.abs(g["edges"][:, 0] - g["edges"][:, 1]) > 1)
    bad = rng.choice(loops, size=min(n_bad, len(loops)), replace=False)
    z[bad, :2] += rng.normal(0, 1.5, (len(bad), 2))
    z[bad, 2] += rng.normal(0, 0.4, len(bad))
    return z, bad


@pytest.mark.parametrize("loss", ["HuberLoss", "CauchyLoss"])
def test_robust_loss_gradient_and_optimum(loss):
    """ceres_solver.cpp:82-94.  The corrected linearisation must be the gradient of 0.5 sum rho(|U r|^2)
    (checked by central differences, both branches of the loss active), the LM run must end at a stationary
    point of that cost, and false loop closures must pull the result less than under the squared loss."""
    g = synth.make_pose_graph(150, 330, seed=31)
    z, bad = _with_outliers(g, 12, seed=32)
    prob = spa.Problem(g["init"], g["edges"], z, g["cov"], loss=loss)
    r, _ = spa._residuals(g["init"], prob.edges[:, 0], prob.edges[:, 1], prob.z, prob.U)
    sq = np.sum(r * r, axis=1)
    assert (sq > 0.49).any() and (sq < 0.49).any()
    _, grad, H = prob.linearize(g["init"])
    g0 = np.abs(grad).max()
    rng = np.random.default_rng(1)
    for k in rng.choice(grad.size, 25, replace=False):
        d = np.zeros(grad.size); d[k] = 1e-6
        num = (prob.cost(prob.plus(g["init"], d)) - prob.cost(prob.plus(g["init"], -d))) / 2e-6
        assert abs(num - grad[k]) <= 1e-5 * max(1.0, abs(grad[k]))
    assert abs(H - H.T).max() < 1e-9
    opt = spa.Options.tight(); opt.loss_function = loss
    x, info = spa.solve(g["init"], g["edges"], z, g["cov"], opt)
    _, grad, _ = spa.Problem(x, g["edges"], z, g["cov"], loss=loss).linearize(x)
    assert np.abs(grad).max() < 1e-6 * g0     # Gauss-Newton on a robustified cost converges linearly
    x2, _ = spa.solve(g["init"], g["edges"], z, g["cov"], spa.Options.tight())
    err_robust = np.abs(x[:, :2] - g["truth"][:, :2]).max()
    err_square = np.abs(x2[:, :2] - g["truth"][:, :2]).max()
    assert err_robust < err_square
