"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement (numpy + scipy.sparse) of the loop-closure solver plugin of slam_toolbox
(hot path B of SURVEY.md section 8): solvers/ceres_solver.cpp + solvers/ceres_utils.h on top of
Ceres Solver's Levenberg-Marquardt trust-region minimiser with SPARSE_NORMAL_CHOLESKY.

PARITY STATUS: **parity unpinned** at the Ceres boundary.  Ceres (libceres-dev, unpinned in
package.xml:32,64; the LocalParameterization API used at ceres_utils.h:50-53 implies Ceres < 2.2,
ROS Humble ships 2.0.0), Eigen and SuiteSparse are not in /root/reference and not installed, and the
reference has no test that pins solver output.  What IS restated from the reference's own files:
  * residual / measurement model        solvers/ceres_utils.h:27-32, 60-68, 84-100
  * information matrix construction     solvers/ceres_solver.cpp:364-376 (+ Matrix3::Inverse Karto.h:2533-2577)
  * gauge (first node constant)         solvers/ceres_solver.cpp:228-241
  * solver options                      solvers/ceres_solver.cpp:157-186
  * LinkInfo::Update                    lib/karto_sdk/include/karto_sdk/Mapper.h:174-188
and pinned against the reference build where the reference code is compilable (LinkInfo::Update and
Matrix3::Inverse known answers in tests/golden/link_info.npz).  The minimiser follows Ceres 2.0's
published algorithm (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
trust_region_step_evaluator.cc: Jacobi scaling 1/(1+sqrt(colnorm^2)) fixed at iteration 0, LM diagonal
clamp, radius update r /= max(1/3, 1-(2rho-1)^3), rejection r /= 2,4,8.., non-monotonic step
acceptance with window 3, parameter/function/gradient tolerance tests).  Independent checks:
noise-free graphs must return the ground truth, and `tight` runs must agree with scipy's generic
least-squares optimum (tests/test_spa_oracle.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

TWO_PI = 2.0 * math.pi


@dataclass
class Options:
    """CeresSolver::Configure (ceres_solver.cpp:157-186) + Ceres defaults."""
    max_num_iterations: int = 50
    function_tolerance: float = 1e-3
    gradient_tolerance: float = 1e-6
    parameter_tolerance: float = 1e-3
    min_relative_decrease: float = 1e-3
    initial_trust_region_radius: float = 1e4
    max_trust_region_radius: float = 1e8
    min_trust_region_radius: float = 1e-16
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    max_num_consecutive_invalid_steps: int = 3
    use_nonmonotonic_steps: bool = True
    max_consecutive_nonmonotonic_steps: int = 3
    jacobi_scaling: bool = True
    loss_function: str = "None"      # ceres_solver.cpp:60-94: "None" | "HuberLoss" | "CauchyLoss"
    loss_scale: float = 0.7          # the scale the plugin hard-wires for both

    @staticmethod
    def tight():
        return Options(max_num_iterations=200, function_tolerance=1e-15, gradient_tolerance=1e-14,
                       parameter_tolerance=1e-14)


def normalize_angle(a):
    """ceres_utils.h:27-32: a - 2*pi*floor((a + pi) / (2*pi)) in [-pi, pi)."""
    return a - TWO_PI * np.floor((a + math.pi) / TWO_PI)


def matrix3_inverse(m):
    """karto::Matrix3::Inverse / InverseFast (Karto.h:2533-2577), cofactor expansion, tolerance 1e-14."""
    m = np.asarray(m, dtype=np.float64).reshape(3, 3)
    inv = np.empty((3, 3))
    inv[0, 0] = m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]
    inv[0, 1] = m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]
    inv[0, 2] = m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]
    inv[1, 0] = m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]
    inv[1, 1] = m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]
    inv[1, 2] = m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]
    inv[2, 0] = m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]
    inv[2, 1] = m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]
    inv[2, 2] = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    det = m[0, 0] * inv[0, 0] + m[0, 1] * inv[1, 0] + m[0, 2] * inv[2, 0]
    if abs(det) <= 1e-14:
        return inv          # assert(false) compiled out in Release: un-normalised cofactors are returned
    return inv * (1.0 / det)


def link_info(pose1, pose2, cov):
    """LinkInfo::Update (Mapper.h:174-188): z = pose2 in the frame of pose1; cov rotated by -theta1."""
    x1, y1, t1 = (float(v) for v in pose1)
    x2, y2, t2 = (float(v) for v in pose2)
    # Transform(rPose1, Pose2()): rotation by (0 - t1); m_Transform = Pose2() - R * pose1 (Karto.h:3003-3024)
    c, s = math.cos(0.0 - t1), math.sin(0.0 - t1)
    if (x1, y1, t1) == (0.0, 0.0, 0.0):
        tx, ty, tth = 0.0, 0.0, 0.0
        c, s = 1.0, 0.0
    elif x1 != 0.0 or y1 != 0.0:
        tx = 0.0 - (c * x1 - s * y1)
        ty = 0.0 - (s * x1 + c * y1)
        tth = 0.0 - t1
    else:
        tx, ty, tth = 0.0, 0.0, 0.0 - t1
    dx = tx + (c * x2 - s * y2)
    dy = ty + (s * x2 + c * y2)
    dth = _karto_normalize(t2 + tth)
    # Matrix3::FromAxisAngle(0, 0, 1, -t1) (Karto.h:2482-2511) and Matrix3 operator* (Karto.h:2634-2647),
    # same operation order: rotationMatrix * rCovariance * rotationMatrix.Transpose()
    cr, sr = math.cos(-t1), math.sin(-t1)
    omc = 1.0 - cr
    R = [[0.0 * omc + cr, 0.0 - sr, 0.0], [0.0 + sr, 0.0 * omc + cr, 0.0], [0.0, 0.0, 1.0 * omc + cr]]
    cv = np.asarray(cov, dtype=np.float64).reshape(3, 3)
    tmp = [[R[r][0] * cv[0][q] + R[r][1] * cv[1][q] + R[r][2] * cv[2][q] for q in range(3)] for r in range(3)]
    out = [[tmp[r][0] * R[q][0] + tmp[r][1] * R[q][1] + tmp[r][2] * R[q][2] for q in range(3)] for r in range(3)]
    return np.array([dx, dy, dth]), np.array(out, dtype=np.float64)


def _karto_normalize(angle):
    """math::NormalizeAngle (Math.h:181-202), range [-pi, pi]."""
    while angle < -math.pi:
        angle += TWO_PI
    while angle > math.pi:
        angle -= TWO_PI
    return angle


def sqrt_information(cov):
    """AddConstraint (ceres_solver.cpp:364-376): Omega = Matrix3::Inverse(cov), symmetrised from its upper
    triangle, U = Omega.llt().matrixU()  (U^T U = Omega)."""
    p = matrix3_inverse(cov)
    info = np.array([[p[0, 0], p[0, 1], p[0, 2]], [p[0, 1], p[1, 1], p[1, 2]], [p[0, 2], p[1, 2], p[2, 2]]])
    return np.linalg.cholesky(info).T


def _residuals(x, ea, eb, z, U):
    """PoseGraph2dErrorTerm (ceres_utils.h:84-100), all edges at once.  x: (N, 3)."""
    xa, xb = x[ea], x[eb]
    c, s = np.cos(xa[:, 2]), np.sin(xa[:, 2])
    dx, dy = xb[:, 0] - xa[:, 0], xb[:, 1] - xa[:, 1]
    raw = np.stack([c * dx + s * dy - z[:, 0], -s * dx + c * dy - z[:, 1],
                    normalize_angle((xb[:, 2] - xa[:, 2]) - z[:, 2])], axis=1)
    return np.einsum("eij,ej->ei", U, raw), (c, s, dx, dy)


def _jacobians(c, s, dx, dy, U):
    """Analytic 3x3 blocks wrt (xa, ya, ta) and (xb, yb, tb); equals Ceres' autodiff to rounding."""
    E = c.shape[0]
    Ja = np.zeros((E, 3, 3))
    Jb = np.zeros((E, 3, 3))
    Ja[:, 0, 0] = -c; Ja[:, 0, 1] = -s; Ja[:, 0, 2] = -s * dx + c * dy
    Ja[:, 1, 0] = s; Ja[:, 1, 1] = -c; Ja[:, 1, 2] = -c * dx - s * dy
    Ja[:, 2, 2] = -1.0
    Jb[:, 0, 0] = c; Jb[:, 0, 1] = s
    Jb[:, 1, 0] = -s; Jb[:, 1, 1] = c
    Jb[:, 2, 2] = 1.0
    return np.einsum("eij,ejk->eik", U, Ja), np.einsum("eij,ejk->eik", U, Jb)


def _loss(sq, kind, a):
    """rho(s), rho'(s) of the plugin's loss functions (ceres/loss_function.cc: HuberLoss::Evaluate,
    CauchyLoss::Evaluate).  Both have rho'' <= 0, so Ceres' Corrector (corrector.cc) reduces to scaling the
    residual and the Jacobian rows by sqrt(rho') -- the alpha term only exists for rho'' > 0."""
    tiny = np.finfo(np.float64).tiny
    if kind in (None, "None"):
        return sq, np.ones_like(sq)
    b = a * a
    if kind == "HuberLoss":
        r = np.sqrt(np.where(sq > b, sq, 1.0))
        return np.where(sq > b, 2.0 * a * r - b, sq), np.where(sq > b, np.maximum(tiny, a / r), 1.0)
    if kind == "CauchyLoss":
        tot = 1.0 + sq * (1.0 / b)
        return b * np.log(tot), np.maximum(tiny, 1.0 / tot)
    raise ValueError(kind)


class Problem:
    """State the plugin keeps: nodes in insertion order, constraints, gauge node."""

    def __init__(self, poses, edges, z, cov, fixed=0, loss="None", loss_scale=0.7):
        self.loss, self.loss_scale = loss, loss_scale
        self.x = np.asarray(poses, dtype=np.float64).copy()
        self.edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        self.z = np.asarray(z, dtype=np.float64).reshape(-1, 3)
        self.U = np.stack([sqrt_information(c) for c in np.asarray(cov).reshape(-1, 9)]) if len(self.edges) else np.zeros((0, 3, 3))
        self.fixed = fixed
        n = self.x.shape[0]
        used = np.zeros(n, dtype=bool)
        used[self.edges.reshape(-1)] = True
        free = used.copy()
        if fixed is not None and fixed >= 0 and used[fixed]:
            free[fixed] = False          # SetParameterBlockConstant only when the blocks exist (ceres_solver.cpp:228-241)
        self.free_nodes = np.flatnonzero(free)
        self.col_of = -np.ones(n, dtype=np.int64)
        self.col_of[self.free_nodes] = np.arange(self.free_nodes.shape[0])
        self.nfree = self.free_nodes.shape[0]

    def cost(self, x):
        r, _ = _residuals(x, self.edges[:, 0], self.edges[:, 1], self.z, self.U)
        if self.loss in (None, "None"):
            return 0.5 * float(np.sum(r * r))
        rho, _ = _loss(np.sum(r * r, axis=1), self.loss, self.loss_scale)
        return 0.5 * float(np.sum(rho))

    def linearize(self, x):
        """cost, gradient g = J^T r and H = J^T J over the free parameters (unscaled)."""
        ea, eb = self.edges[:, 0], self.edges[:, 1]
        r, (c, s, dx, dy) = _residuals(x, ea, eb, self.z, self.U)
        Ja, Jb = _jacobians(c, s, dx, dy, self.U)
        if self.loss in (None, "None"):
            cost = 0.5 * float(np.sum(r * r))
        else:
            rho, rho1 = _loss(np.sum(r * r, axis=1), self.loss, self.loss_scale)
            cost = 0.5 * float(np.sum(rho))
            w = np.sqrt(rho1)
            r = r * w[:, None]
            Ja = Ja * w[:, None, None]
            Jb = Jb * w[:, None, None]
        n3 = 3 * self.nfree
        g = np.zeros(n3)
        ca, cb = self.col_of[ea], self.col_of[eb]
        ga = np.einsum("eji,ej->ei", Ja, r)
        gb = np.einsum("eji,ej->ei", Jb, r)
        ma, mb = ca >= 0, cb >= 0
        np.add.at(g, (3 * ca[ma, None] + np.arange(3)).reshape(-1), ga[ma].reshape(-1))
        np.add.at(g, (3 * cb[mb, None] + np.arange(3)).reshape(-1), gb[mb].reshape(-1))
        rows, cols, vals = [], [], []

        def add(ci, cj, blocks, mask):
            if not mask.any():
                return
            ii = 3 * ci[mask, None, None] + np.arange(3)[None, :, None] + np.zeros((1, 1, 3), dtype=np.int64)
            jj = 3 * cj[mask, None, None] + np.arange(3)[None, None, :] + np.zeros((1, 3, 1), dtype=np.int64)
            rows.append(ii.reshape(-1)); cols.append(jj.reshape(-1)); vals.append(blocks[mask].reshape(-1))
        add(ca, ca, np.einsum("eki,ekj->eij", Ja, Ja), ma)
        add(cb, cb, np.einsum("eki,ekj->eij", Jb, Jb), mb)
        both = ma & mb
        Hab = np.einsum("eki,ekj->eij", Ja, Jb)
        add(ca, cb, Hab, both)
        add(cb, ca, np.transpose(Hab, (0, 2, 1)), both)
        if rows:
            H = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n3, n3)).tocsc()
        else:
            H = sp.csc_matrix((n3, n3))
        return cost, g, H

    def plus(self, x, delta):
        """x (+) delta over the free parameters; angles through AngleLocalParameterization (ceres_utils.h:38-55)."""
        out = x.copy()
        d = delta.reshape(-1, 3)
        out[self.free_nodes, 0] += d[:, 0]
        out[self.free_nodes, 1] += d[:, 1]
        out[self.free_nodes, 2] = normalize_angle(out[self.free_nodes, 2] + d[:, 2])
        return out

    def free_vector(self, x):
        return x[self.free_nodes].reshape(-1)


def solve(poses, edges, z, cov, options: Options = None, fixed=0):
    """CeresSolver::Compute (ceres_solver.cpp:214-269) with Ceres' trust-region LM restated.
    Returns (poses (N,3), info dict)."""
    opt = options or Options()
    prob = Problem(poses, edges, z, cov, fixed, opt.loss_function, opt.loss_scale)
    info = dict(iterations=0, successful_steps=0, termination="NO_CONVERGENCE", usable=True, message="", costs=[])
    x = prob.x.copy()
    if prob.nfree == 0 or len(prob.edges) == 0:
        info.update(termination="CONVERGENCE", initial_cost=0.0, final_cost=0.0, message="no free parameters")
        return x, info

    # ---- iteration zero (TrustRegionMinimizer::IterationZero / EvaluateGradientAndJacobian) ----
    x_cost, g, H = prob.linearize(x)
    info["initial_cost"] = x_cost
    if opt.jacobi_scaling:
        scale = 1.0 / (1.0 + np.sqrt(H.diagonal()))
    else:
        scale = np.ones(3 * prob.nfree)

    def grad_norms(xc, gc):
        # projected gradient step through Plus (trust_region_minimizer.cc: EvaluateGradientAndJacobian)
        xp = prob.plus(xc, -gc)
        d = prob.free_vector(xc) - prob.free_vector(xp)
        return float(np.max(np.abs(d))), float(np.linalg.norm(d))

    gmax, gnorm = grad_norms(x, g)
    x_norm = float(np.linalg.norm(prob.free_vector(x)))
    best_x, minimum_cost = x.copy(), x_cost
    # TrustRegionStepEvaluator
    max_nonmono = opt.max_consecutive_nonmonotonic_steps if opt.use_nonmonotonic_steps else 0
    ev = dict(minimum=x_cost, current=x_cost, reference=x_cost, candidate=x_cost, acc_ref=0.0, acc_cand=0.0, nonmono=0)
    # LevenbergMarquardtStrategy
    radius, decrease_factor, reuse_diagonal = opt.initial_trust_region_radius, 2.0, False
    diagonal = None
    num_invalid = 0
    info["costs"].append(x_cost)
    info["successful_steps"] = 1
    iteration = 0
    step_successful = True

    while True:
        # FinalizeIterationAndCheckIfMinimizerCanContinue
        if iteration >= opt.max_num_iterations:
            info.update(termination="NO_CONVERGENCE", message="Maximum number of iterations reached.")
            break
        if step_successful and gmax <= opt.gradient_tolerance:
            info.update(termination="CONVERGENCE", message="Gradient tolerance reached.")
            break
        if radius < opt.min_trust_region_radius:
            info.update(termination="CONVERGENCE", message="Minimum trust region radius reached.")
            break
        iteration += 1
        step_successful = False

        # ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep) ----
        S = sp.diags(scale)
        Hs = (S @ H @ S).tocsc()
        gs = scale * g
        if not reuse_diagonal or diagonal is None:
            diagonal = np.clip(Hs.diagonal(), opt.min_lm_diagonal, opt.max_lm_diagonal)
        lm = diagonal / radius
        step_valid = True
        try:
            lu = spla.splu((Hs + sp.diags(lm)).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                           options=dict(SymmetricMode=True))
            y = lu.solve(gs)
            if not np.all(np.isfinite(y)):
                step_valid = False
        except RuntimeError:
            step_valid = False
        reuse_diagonal = True
        if step_valid:
            step = -y
            # model_cost_change = -(J s)^T (r + J s / 2) = -(s.g + s^T H s / 2)
            model_cost_change = -(float(step @ gs) + 0.5 * float(step @ (Hs @ step)))
            step_valid = model_cost_change > 0.0
        if not step_valid:
            num_invalid += 1
            if num_invalid >= opt.max_num_consecutive_invalid_steps:
                info.update(termination="FAILURE", usable=False,
                            message="Number of consecutive invalid steps more than max_num_consecutive_invalid_steps")
                break
            radius = radius / decrease_factor      # StepIsInvalid -> StepRejected(0)
            decrease_factor *= 2.0
            reuse_diagonal = True
            continue
        num_invalid = 0
        delta = step * scale

        # ---- candidate, tolerances ----
        cand = prob.plus(x, delta)
        cand_cost = prob.cost(cand)
        step_norm = float(np.linalg.norm(prob.free_vector(x) - prob.free_vector(cand)))
        if step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance):
            info.update(termination="CONVERGENCE", message="Parameter tolerance reached.")
            break
        cost_change = x_cost - cand_cost
        if abs(cost_change) <= opt.function_tolerance * x_cost:
            info.update(termination="CONVERGENCE", message="Function tolerance reached.")
            break
        # ---- IsStepSuccessful (TrustRegionStepEvaluator::StepQuality) ----
        rel = (ev["current"] - cand_cost) / model_cost_change
        hist = (ev["reference"] - cand_cost) / (ev["acc_ref"] + model_cost_change)
        quality = max(rel, hist)
        if quality > opt.min_relative_decrease:
            # HandleSuccessfulStep
            x = cand
            x_norm = float(np.linalg.norm(prob.free_vector(x)))
            x_cost, g, H = prob.linearize(x)
            gmax, gnorm = grad_norms(x, g)
            step_successful = True
            info["successful_steps"] += 1
            radius = radius / max(1.0 / 3.0, 1.0 - (2.0 * quality - 1.0) ** 3)
            radius = min(opt.max_trust_region_radius, radius)
            decrease_factor = 2.0
            reuse_diagonal = False
            # StepAccepted
            ev["current"] = cand_cost
            ev["acc_cand"] += model_cost_change
            ev["acc_ref"] += model_cost_change
            if ev["current"] < ev["minimum"]:
                ev["minimum"] = ev["current"]
                ev["nonmono"] = 0
                ev["candidate"] = ev["current"]
                ev["acc_cand"] = 0.0
            else:
                ev["nonmono"] += 1
                if ev["current"] > ev["candidate"]:
                    ev["candidate"] = ev["current"]
                    ev["acc_cand"] = 0.0
            if ev["nonmono"] == max_nonmono:
                ev["reference"] = ev["candidate"]
                ev["acc_ref"] = ev["acc_cand"]
            if x_cost < minimum_cost:
                minimum_cost = x_cost
                best_x = x.copy()
            info["costs"].append(x_cost)
        else:
            radius = radius / decrease_factor       # StepRejected
            decrease_factor *= 2.0
            reuse_diagonal = True

    info["iterations"] = iteration
    info["final_cost"] = minimum_cost
    if not info["usable"]:
        return prob.x.copy(), info          # ceres_solver.cpp:249-254: keep the old state
    return best_x, info
