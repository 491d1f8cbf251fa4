"""CPU: the data-parallel formulation of FindValidPoints that k_find_valid_par runs on the device (next() per reading,
pointer doubling for the readings the walk visits, one side-of-line test per visited trigger, "first trigger after me"
per reading -- csrc/matcher_kernels.hip) restated in numpy, against the sequential restatement of the reference
(oracle/karto_oracle.c::ko_find_valid_points, Mapper.cpp:1113-1164).  The kernel itself is covered by the GPU parity
tests; this pins the ALGORITHM, including the corners the synthetic scans rarely produce."""
import ctypes as C

import numpy as np
import pytest

from common import LASER


def flags_by_pointer_doubling(points, viewpoint):
    """-> uint8 flags, 1 = the reading is emitted; the same steps, in the same order, as the kernel"""
    n = points.shape[0]
    out = np.zeros(n, dtype=np.uint8)
    valid = ~np.isnan(points[:, 0]) & ~np.isnan(points[:, 1])
    if not valid.any():
        return out
    pos0 = int(np.argmax(valid))
    # next(i): first reading after i more than 0.1 m from reading i (never for a NaN reading)
    nxt = np.full(n + 1, n, dtype=np.int64)
    for i in range(n):
        if not valid[i]:
            continue
        with np.errstate(invalid="ignore"):
            dx = points[i, 0] - points[i + 1:, 0]
            dy = points[i, 1] - points[i + 1:, 1]
            hit = dx * dx + dy * dy > 0.1 * 0.1
        if hit.any():
            nxt[i] = i + 1 + int(np.argmax(hit))
    # readings the walk visits, by pointer doubling
    reach = np.zeros(n + 1, dtype=bool)
    reach[pos0] = True
    cur = nxt.copy()
    span = 1
    while span < n:
        src = np.flatnonzero(reach[:n])
        tgt = cur[src]
        reach[tgt[tgt < n]] = True
        cur = np.where(cur < n, cur[np.minimum(cur, n)], n)
        span <<= 1
    # side of the line viewpoint -> anchor for every visited trigger
    keep = np.zeros(n, dtype=np.uint8)
    vx, vy = viewpoint
    for i in np.flatnonzero(reach[:n]):
        j = int(nxt[i])
        if j < n:
            fx, fy = points[i]
            cx, cy = points[j]
            with np.errstate(invalid="ignore"):          # +inf beams: inf - inf = NaN, compares false like in C
                a = vy - fy
                b = fx - vx
                c = fy * vx - fx * vy
                ss = cx * a + cy * b + c
            keep[j] = 0 if ss < 0.0 else 1
    # a reading belongs to the run that ends at the first trigger after it; the tail after the last trigger is dropped
    trig = reach[:n].copy()
    trig[pos0] = False
    later = -1
    for i in range(n - 1, -1, -1):
        out[i] = keep[later] if later >= 0 else 0
        if trig[i]:
            later = i
    return out


def _scan_points(ranges, pose):
    ang = pose[2] + LASER.min_angle + np.arange(ranges.size) * LASER.ang_res
    return np.stack([pose[0] + ranges * np.cos(ang), pose[1] + ranges * np.sin(ang)], axis=1)


def _cases():
    rng = np.random.default_rng(5)
    n = 1081
    pose = np.array([1.0, -2.0, 0.3])
    # (a) a room: smooth ranges, noise, NaN / inf beams
    base = 4.0 + 2.0 * np.sin(np.linspace(0, 9, n)) + rng.normal(0, 0.01, n)
    r = base.copy()
    r[rng.uniform(size=n) < 0.02] = np.nan
    r[rng.uniform(size=n) < 0.02] = np.inf
    yield _scan_points(r, pose), pose[:2]
    # (b) long beams: every reading is a trigger
    yield _scan_points(np.full(n, 28.0) + rng.normal(0, 0.3, n), pose), pose[:2]
    # (c) short beams: dozens of readings between triggers, with a run of NaN in the middle
    r = np.full(n, 0.6) + rng.normal(0, 0.002, n)
    r[300:520] = np.nan
    yield _scan_points(r, pose), pose[:2]
    # (d) starts with NaN readings, viewpoint away from the sensor
    r = base.copy()
    r[:37] = np.nan
    yield _scan_points(r, pose), np.array([3.0, 3.0])
    # (e) all NaN, a single valid reading, a ragged short scan
    yield np.full((n, 2), np.nan), pose[:2]
    p = np.full((50, 2), np.nan); p[17] = (1.0, 2.0)
    yield p, pose[:2]
    yield _scan_points(base[:63], pose), pose[:2]
    # (f) jagged: alternating near / far returns (occlusion edges flip the side-of-line sign)
    r = np.where(np.arange(n) % 7 < 3, 2.0, 9.0) + rng.normal(0, 0.05, n)
    yield _scan_points(r, pose), pose[:2]


@pytest.mark.parametrize("case", range(8))
def test_pointer_doubling_equals_the_sequential_state_machine(oracle_lib, case):
    from oracle import karto
    points, view = list(_cases())[case]
    flags = flags_by_pointer_doubling(points, view)
    # FindValidPoints only reads the scan's points
    scan = karto.Scan(np.ones(points.shape[0]), np.zeros(3), points=np.ascontiguousarray(points))
    cs = scan.c()
    out = np.zeros((scan.n, 2))
    view_c = np.ascontiguousarray(view, dtype=np.float64)
    n_out = karto.lib().ko_find_valid_points(C.byref(cs), view_c, out)
    want = out[:n_out]
    got = points[flags.astype(bool)]
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), np.asarray(want).view(np.uint64))


def _range_cases():
    rng = np.random.default_rng(9)
    n = LASER.n_beams
    base = 4.0 + 2.0 * np.sin(np.linspace(0, 9, n)) + rng.normal(0, 0.01, n)
    r0 = base.copy()
    r0[rng.uniform(size=n) < 0.03] = np.nan
    r0[rng.uniform(size=n) < 0.03] = np.inf
    r1 = np.where(np.arange(n) % 9 < 4, 1.5, 12.0) + rng.normal(0, 0.05, n)
    r2 = np.full(n, 0.7) + rng.normal(0, 0.002, n)
    r2[:60] = np.nan
    r2[400:640] = np.inf
    return [r0, r1, r2]


@pytest.mark.parametrize("case", range(3))
def test_pointer_doubling_against_the_reference_itself(case):
    """the same flags from the reference's own ScanMatcher::FindValidPoints (oracle/_ref, dev container only) on scans the
    reference builds from ranges and pose"""
    from common import OFFLINE_PARAMS, PRESETS
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref.init_laser(LASER)
    pose = np.array([2.0, 1.0, -0.4])
    rs = ref.RefScan(_range_cases()[case], pose)
    points = rs.points()
    assert points.shape[0] == LASER.n_beams
    view = pose[:2] + np.array([0.3, -0.2])
    rm = ref.RefMatcher(*PRESETS["S"]["create"], OFFLINE_PARAMS)
    want = rm.find_valid_points(rs, view)
    got = points[flags_by_pointer_doubling(points, view).astype(bool)]
    assert got.shape == want.shape and got.shape[0] > 0
    assert np.array_equal(got.view(np.uint64), np.ascontiguousarray(want).view(np.uint64))
