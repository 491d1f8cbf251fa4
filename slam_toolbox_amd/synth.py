"""Deterministic synthetic world + 1081-beam laser scans.

Shared by the tests, the golden-fixture generator and bench.py.  Pure numpy; no reference
or oracle code is involved.  World: 60 m x 40 m warehouse box, aisles of 1 m x 8 m rack
rectangles at 3 m pitch, random 0.3 m square pillars.  Laser: Hokuyo UTM-30LX geometry
(-135..+135 deg at 0.25 deg -> 1081 beams; karto preset Karto.h:4191-4207), min range 0.1,
max range 30, range threshold 20 (config/mapper_params_offline.yaml:24).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

N_BEAMS = 1081
MIN_ANGLE = math.radians(-135.0)
MAX_ANGLE = math.radians(135.0)
ANG_RES = math.radians(0.25)
MIN_RANGE = 0.1
MAX_RANGE = 30.0
RANGE_THRESHOLD = 20.0


@dataclass
class Laser:
    n_beams: int = N_BEAMS
    min_angle: float = MIN_ANGLE
    max_angle: float = MAX_ANGLE
    ang_res: float = ANG_RES
    min_range: float = MIN_RANGE
    max_range: float = MAX_RANGE
    range_threshold: float = RANGE_THRESHOLD
    offset: tuple = (0.0, 0.0, 0.0)        # LaserRangeFinder::GetOffsetPose: the sensor on the robot (x, y, heading)


def _rect(x0, y0, x1, y1):
    return [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]


def make_world(seed: int = 12345, n_pillars: int = 40) -> np.ndarray:
    """Returns line segments (S, 4) = x0, y0, x1, y1."""
    rng = np.random.default_rng(seed)
    segs = _rect(0.0, 0.0, 60.0, 40.0)
    # rack rows: 1 m wide, 8 m long, 3 m pitch in x, two bands in y
    x = 4.0
    while x + 1.0 < 56.0:
        for y0 in (6.0, 18.0, 28.0):
            segs += _rect(x, y0, x + 1.0, y0 + 8.0)
        x += 4.0
    for _ in range(n_pillars):
        px = rng.uniform(1.0, 59.0)
        py = rng.uniform(1.0, 39.0)
        segs += _rect(px, py, px + 0.3, py + 0.3)
    return np.asarray(segs, dtype=np.float64)


def raycast(world: np.ndarray, pose, laser: Laser = Laser()) -> np.ndarray:
    """Exact ray/segment intersection ranges (inf when nothing is hit)."""
    x, y, th = pose
    ang = th + laser.min_angle + np.arange(laser.n_beams) * laser.ang_res
    dx = np.cos(ang)[:, None]
    dy = np.sin(ang)[:, None]
    x0, y0, x1, y1 = (world[:, i][None, :] for i in range(4))
    ex, ey = x1 - x0, y1 - y0
    denom = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((x0 - x) * ey - (y0 - y) * ex) / denom
        u = ((x0 - x) * dy - (y0 - y) * dx) / denom
    ok = (np.abs(denom) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    return t.min(axis=1)


def make_scan(world, pose, rng, laser: Laser = Laser(), noise=0.01, p_inf=0.01, p_nan=0.005):
    """Noisy ranges with a few +inf / NaN beams (exercises INVALID_SCAN, Karto.h:6869-6875)."""
    r = raycast(world, pose, laser)
    r = r + rng.normal(0.0, noise, size=r.shape)
    r = np.where(r > laser.max_range, laser.max_range, r)
    r = np.maximum(r, 0.02)
    flags = rng.uniform(size=r.shape)
    r = np.where(flags < p_inf, np.inf, r)
    r = np.where((flags >= p_inf) & (flags < p_inf + p_nan), np.nan, r)
    return r


def inside_obstacle(world, x, y, margin=0.6) -> bool:
    if x < margin or y < margin or x > 60.0 - margin or y > 40.0 - margin:
        return True
    # rectangles are stored as 4 consecutive segments
    for k in range(4, world.shape[0], 4):
        xs = world[k:k + 4, [0, 2]]
        ys = world[k:k + 4, [1, 3]]
        if xs.min() - margin <= x <= xs.max() + margin and ys.min() - margin <= y <= ys.max() + margin:
            return True
    return False


def trajectory(n_nodes: int, spacing: float = 0.5, seed: int = 12345):
    """Boustrophedon through the aisles: true poses (n,3) and drifting odometry (n,3)."""
    rng = np.random.default_rng(seed + 1)
    # way-points along the aisle centre lines (x = 2.5 + 4k), sweeping y 3..37
    pts = []
    k = 0
    xc = 2.5
    while xc < 58.0:
        ys = (3.0, 37.0) if k % 2 == 0 else (37.0, 3.0)
        pts.append((xc, ys[0]))
        pts.append((xc, ys[1]))
        xc += 4.0
        k += 1
    poses = []
    i = 0
    cur = np.array(pts[0], dtype=np.float64)
    tgt = 1
    heading = math.atan2(pts[1][1] - pts[0][1], pts[1][0] - pts[0][0])
    direction = 1
    while len(poses) < n_nodes:
        poses.append((cur[0], cur[1], heading))
        d = np.array(pts[tgt]) - cur
        dist = float(np.hypot(*d))
        if dist < spacing:
            cur = np.array(pts[tgt], dtype=np.float64)
            nxt = tgt + direction
            if nxt >= len(pts) or nxt < 0:
                direction = -direction
                nxt = tgt + direction
            tgt = nxt
            d = np.array(pts[tgt]) - cur
            heading = math.atan2(d[1], d[0])
        else:
            cur = cur + d / dist * spacing
            heading = math.atan2(d[1], d[0])
        i += 1
    truth = np.asarray(poses, dtype=np.float64)
    # odometry = truth composed with a random-walk drift
    odom = np.zeros_like(truth)
    odom[0] = truth[0]
    for j in range(1, n_nodes):
        # true relative motion in the previous true frame
        c, s = math.cos(truth[j - 1, 2]), math.sin(truth[j - 1, 2])
        dxw, dyw = truth[j, 0] - truth[j - 1, 0], truth[j, 1] - truth[j - 1, 1]
        dxl = c * dxw + s * dyw + rng.normal(0.0, 0.02)
        dyl = -s * dxw + c * dyw + rng.normal(0.0, 0.02)
        dth = truth[j, 2] - truth[j - 1, 2]
        dth = (dth + math.pi) % (2 * math.pi) - math.pi + rng.normal(0.0, math.radians(0.5))
        c2, s2 = math.cos(odom[j - 1, 2]), math.sin(odom[j - 1, 2])
        odom[j, 0] = odom[j - 1, 0] + c2 * dxl - s2 * dyl
        odom[j, 1] = odom[j - 1, 1] + s2 * dxl + c2 * dyl
        odom[j, 2] = odom[j - 1, 2] + dth
        odom[j, 2] = (odom[j, 2] + math.pi) % (2 * math.pi) - math.pi
    return truth, odom


def scan_points(ranges: np.ndarray, sensor_pose, laser: Laser = Laser()) -> np.ndarray:
    """Unfiltered world points the way LocalizedRangeScan::Update makes them
    (Karto.h:5644-5704): angle = heading + minAngle + i*angRes; pt = pose + r*(cos, sin).
    Uses libm cos/sin through python's math module so it is bit-identical to the C side."""
    x, y, th = (float(v) for v in sensor_pose)
    out = np.empty((ranges.shape[0], 2), dtype=np.float64)
    for i in range(ranges.shape[0]):
        a = th + laser.min_angle + i * laser.ang_res
        r = float(ranges[i])
        out[i, 0] = x + (r * math.cos(a))
        out[i, 1] = y + (r * math.sin(a))
    return out


def make_pose_graph(n_nodes: int, n_edges: int, seed: int = 12345, max_link_dist: float = 3.0):
    """BASELINE config 4: n_nodes poses, n_nodes-1 odometry edges + extra edges between nodes
    closer than max_link_dist; z = true relative pose (+) noise (1 cm, 0.2 deg);
    Sigma = R diag(1e-3, 1e-3, 4e-4) R^T with random in-plane rotation; initial = drifted odometry.
    Returns dict(truth, init, edges (E,2) int32, z (E,3), cov (E,9))."""
    truth, odom = trajectory(n_nodes, 0.5, seed)
    rng = np.random.default_rng(seed + 2)
    pairs = [(i, i + 1) for i in range(n_nodes - 1)]
    have = set(pairs)
    # spatial hash for near pairs
    cell = max_link_dist
    buckets = {}
    for i in range(n_nodes):
        key = (int(truth[i, 0] // cell), int(truth[i, 1] // cell))
        buckets.setdefault(key, []).append(i)
    cand = []
    for (cx, cy), members in buckets.items():
        for ox in (-1, 0, 1):
            for oy in (-1, 0, 1):
                other = buckets.get((cx + ox, cy + oy))
                if not other:
                    continue
                for i in members:
                    for j in other:
                        if j > i + 1 and math.hypot(truth[i, 0] - truth[j, 0], truth[i, 1] - truth[j, 1]) < max_link_dist:
                            cand.append((i, j))
    cand = sorted(set(cand))
    need = n_edges - len(pairs)
    if need > 0:
        if need > len(cand):
            raise ValueError(f"only {len(cand)} near pairs available, need {need}")
        idx = rng.choice(len(cand), size=need, replace=False)
        idx.sort()
        pairs += [cand[k] for k in idx]
    edges = np.asarray(pairs, dtype=np.int32)
    E = edges.shape[0]
    z = np.zeros((E, 3))
    cov = np.zeros((E, 9))
    for e, (a, b) in enumerate(pairs):
        c, s = math.cos(truth[a, 2]), math.sin(truth[a, 2])
        dxw, dyw = truth[b, 0] - truth[a, 0], truth[b, 1] - truth[a, 1]
        z[e, 0] = c * dxw + s * dyw + rng.normal(0.0, 0.01)
        z[e, 1] = -s * dxw + c * dyw + rng.normal(0.0, 0.01)
        dth = truth[b, 2] - truth[a, 2] + rng.normal(0.0, math.radians(0.2))
        z[e, 2] = (dth + math.pi) % (2 * math.pi) - math.pi
        phi = rng.uniform(0, 2 * math.pi)
        R = np.array([[math.cos(phi), -math.sin(phi), 0.0], [math.sin(phi), math.cos(phi), 0.0], [0, 0, 1.0]])
        S = R @ np.diag([1e-3, 1e-3, 4e-4]) @ R.T
        cov[e] = (0.5 * (S + S.T)).reshape(9)
    return {"truth": truth, "init": odom.copy(), "edges": edges, "z": z, "cov": cov}


def trajectory_laps(n_nodes: int, spacing: float = 0.5, seed: int = 12345, aisles=(0, 1), drift_xy=0.02,
                    drift_theta_deg=0.5):
    """Closed circuit driven lap after lap ("multiple passes", BASELINE config 5): up aisle aisles[0], across the
    top, down aisle aisles[1], across the bottom, again.  Every lap after the first revisits the poses of the one
    before within centimetres, so loop closure fires with the shipped loop_search_maximum_distance of 3 m
    (config/mapper_params_offline.yaml:40).  Returns truth (n,3) and drifting odometry (n,3) like trajectory()."""
    rng = np.random.default_rng(seed + 7)
    xa, xb = 2.5 + 4.0 * aisles[0], 2.5 + 4.0 * aisles[1]
    corners = [(xa, 3.0), (xa, 37.0), (xb, 37.0), (xb, 3.0)]
    poses = []
    cur = np.array(corners[0], dtype=np.float64)
    tgt = 1
    heading = math.atan2(corners[1][1] - corners[0][1], corners[1][0] - corners[0][0])
    while len(poses) < n_nodes:
        poses.append((cur[0], cur[1], heading))
        d = np.array(corners[tgt]) - cur
        dist = float(np.hypot(*d))
        if dist < spacing:
            cur = np.array(corners[tgt], dtype=np.float64)
            tgt = (tgt + 1) % 4
            d = np.array(corners[tgt]) - cur
            heading = math.atan2(d[1], d[0])
        else:
            cur = cur + d / dist * spacing
            heading = math.atan2(d[1], d[0])
    truth = np.asarray(poses, dtype=np.float64)
    odom = np.zeros_like(truth)
    odom[0] = truth[0]
    for j in range(1, n_nodes):
        c, s = math.cos(truth[j - 1, 2]), math.sin(truth[j - 1, 2])
        dxw, dyw = truth[j, 0] - truth[j - 1, 0], truth[j, 1] - truth[j - 1, 1]
        dxl = c * dxw + s * dyw + rng.normal(0.0, drift_xy)
        dyl = -s * dxw + c * dyw + rng.normal(0.0, drift_xy)
        dth = truth[j, 2] - truth[j - 1, 2]
        dth = (dth + math.pi) % (2 * math.pi) - math.pi + rng.normal(0.0, math.radians(drift_theta_deg))
        c2, s2 = math.cos(odom[j - 1, 2]), math.sin(odom[j - 1, 2])
        odom[j, 0] = odom[j - 1, 0] + c2 * dxl - s2 * dyl
        odom[j, 1] = odom[j - 1, 1] + s2 * dxl + c2 * dyl
        odom[j, 2] = (odom[j - 1, 2] + dth + math.pi) % (2 * math.pi) - math.pi
    return truth, odom


def loop_batch(n_pairs: int = 256, seed: int = 99, n_traj: int = 2000, world_seed: int = 12345):
    """BASELINE config 2 ("loop-closure batch: 256 candidate pairs on a 2k-node graph"): n_pairs DISTINCT (query scan,
    candidate chain of 10..40 consecutive scans) pairs on the 2000-node warehouse trajectory; the chain is centred on
    the trajectory node nearest to the query among those at least 80 nodes away along the graph, the query pose is
    perturbed by up to (0.15 m, 0.1 m, 0.06 rad).  Returns dict(ranges {node: (P,)}, truth (n,3),
    pairs [(query node, query pose (3,), [chain nodes])]) -- raw data; callers wrap it into their scan type."""
    world = make_world(world_seed)
    truth, _ = trajectory(n_traj)
    rng = np.random.default_rng(seed)
    ranges = {}

    def need(i):
        if i not in ranges:
            ranges[i] = make_scan(world, truth[i], rng)
    pairs = []
    for k in range(n_pairs):
        q = 150 + (53 * k) % (n_traj - 300)
        d = np.hypot(truth[:, 0] - truth[q, 0], truth[:, 1] - truth[q, 1])
        d[max(0, q - 80): q + 80] = 1e9
        j = int(np.argmin(d))
        length = 10 + (7 * k) % 31
        lo = max(0, min(n_traj - length, j - length // 2))
        chain = list(range(lo, lo + length))
        for i in chain + [q]:
            need(i)
        pose = truth[q] + np.array([0.15 * math.sin(k), -0.1 * math.cos(k), 0.03 * ((k % 5) - 2)])
        pairs.append((q, pose, chain))
    return {"ranges": ranges, "truth": truth, "pairs": pairs}
