"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement of the loop-candidate enumeration of karto::MapperGraph:
  * FindNearLinkedScans   Mapper.cpp:1795-1806  = BreadthFirstTraversal::TraverseForVertices (Mapper.cpp:1263-1297)
                          with NearScanVisitor (Mapper.cpp:1311-1333) over Vertex::GetAdjacentVertices
                          (Mapper.h:338-361)
  * FindPossibleLoopClosure  Mapper.cpp:1960-2010, called repeatedly like TryCloseLoop does (Mapper.cpp:1500-1560)
PINNED against the reference build: tests/golden/loop_candidates.npz (tests/golden/make_golden_loops.py) holds what
the reference's own functions returned for every scan of a 223-node graph built by the reference Mapper.

Graph store: ref_xy (N, 2) = GetReferencePose(use_scan_barycenter) positions in scan-list order (NULL scans are
simply absent from the list, they are skipped by the reference, Mapper.cpp:1980-1982); adjacency in CSR form,
neighbours in Vertex::GetAdjacentVertices order."""
from collections import deque

import numpy as np

KT_TOLERANCE = 1e-06          # Math.h:41


def squared_distance(a, b):
    """Vector2::SquaredDistance: Square(x - x') + Square(y - y')."""
    dx = float(a[0]) - float(b[0])
    dy = float(a[1]) - float(b[1])
    return dx * dx + dy * dy


def near_linked_scans(q, ref_xy, adj_ptr, adj_idx, max_distance):
    """Visited-and-valid vertices in BFS order (the start vertex is visited like any other)."""
    lim = max_distance * max_distance - KT_TOLERANCE        # NearScanVisitor::Visit, Mapper.cpp:1326-1327
    centre = ref_xy[q]
    to_visit = deque([q])
    seen = {q}
    valid = []
    while to_visit:
        v = to_visit.popleft()
        if squared_distance(ref_xy[v], centre) <= lim:
            valid.append(v)
            for w in adj_idx[adj_ptr[v]: adj_ptr[v + 1]]:
                w = int(w)
                if w not in seen:
                    seen.add(w)
                    to_visit.append(w)
    return valid


def find_possible_loop_closures(q, ref_xy, adj_ptr, adj_idx, max_distance, min_chain_size):
    """All chains successive FindPossibleLoopClosure(pScan, sensor, rStartNum) calls return, as (first, last)
    scan indices (chains are runs of consecutive scans)."""
    linked = set(near_linked_scans(q, ref_xy, adj_ptr, adj_idx, max_distance))
    lim = max_distance * max_distance + KT_TOLERANCE        # Mapper.cpp:1988-1990
    n = ref_xy.shape[0]
    pose = ref_xy[q]
    out = []
    start = 0
    while True:
        chain = []
        returned = False
        while start < n:
            if squared_distance(ref_xy[start], pose) < lim:
                if start in linked:
                    chain = []                               # a linked scan cannot be in the chain
                else:
                    chain.append(start)
            else:
                if len(chain) >= min_chain_size:
                    returned = True                          # rStartNum is NOT advanced on this return
                    break
                chain = []
            start += 1
        if not chain:
            break                                            # TryCloseLoop stops at the first empty chain
        out.append((chain[0], chain[-1]))
        if not returned:
            break                                            # end of the scan list: the next call returns empty
    return out


def find_near_chains(q, ref_xy, adj_ptr, adj_idx, link_max_distance):
    """MapperGraph::FindNearChains (Mapper.cpp:1683-1793) as (first, last) runs, in the reference's order."""
    lim = link_max_distance * link_max_distance + KT_TOLERANCE          # Mapper.cpp:1735-1737
    n = ref_xy.shape[0]
    pose = ref_xy[q]
    processed = set()
    out = []
    for near in near_linked_scans(q, ref_xy, adj_ptr, adj_idx, link_max_distance):
        if near == q or near in processed:
            continue
        processed.add(near)
        valid = True
        first = last = near
        c = near - 1
        while c >= 0:
            if c == q:
                valid = False
            if squared_distance(pose, ref_xy[c]) < lim:
                first = c
                processed.add(c)
            else:
                break
            c -= 1
        c = near + 1
        while c < n:
            if c == q:
                valid = False
            if squared_distance(pose, ref_xy[c]) < lim:
                last = c
                processed.add(c)
            else:
                break
            c += 1
        if valid:
            out.append((first, last))
    return out


def closest_scan_to_pose(scans, ref_xy, pose_xy):
    """MapperGraph::GetClosestScanToPose (Mapper.cpp:1563-1582): the first strictly smaller distance wins."""
    best, best_d = -1, float("inf")
    for s in scans:
        d = squared_distance(pose_xy, ref_xy[s])
        if d < best_d:
            best, best_d = int(s), d
    return best


def compute_weighted_mean(means, covariances):
    """MapperGraph::ComputeWeightedMean (Mapper.cpp:1914-1958) with Matrix3::Inverse (Karto.h:2533-2577), the 3x3
    product (Karto.h:2634-2647) and matrix * pose (Karto.h:2654-2666) in the reference's operation order."""
    import math

    def inverse(m):
        inv = [m[4] * m[8] - m[5] * m[7], m[2] * m[7] - m[1] * m[8], m[1] * m[5] - m[2] * m[4],
               m[5] * m[6] - m[3] * m[8], m[0] * m[8] - m[2] * m[6], m[2] * m[3] - m[0] * m[5],
               m[3] * m[7] - m[4] * m[6], m[1] * m[6] - m[0] * m[7], m[0] * m[4] - m[1] * m[3]]
        det = m[0] * inv[0] + m[1] * inv[3] + m[2] * inv[6]
        if abs(det) <= 1e-14:
            return inv
        inv_det = 1.0 / det
        return [v * inv_det for v in inv]
    means = np.asarray(means, dtype=np.float64).reshape(-1, 3)
    covs = np.asarray(covariances, dtype=np.float64).reshape(-1, 9)
    inverses = [inverse([float(v) for v in c]) for c in covs]
    total = [0.0] * 9
    for inv in inverses:
        total = [a + b for a, b in zip(total, inv)]
    inv_sum = inverse(total)
    ax = ay = tx = ty = 0.0
    for p, inv in zip(means, inverses):
        x, y, h = float(p[0]), float(p[1]), float(p[2])
        tx += math.cos(h)
        ty += math.sin(h)
        w = [inv_sum[3 * r] * inv[c] + inv_sum[3 * r + 1] * inv[3 + c] + inv_sum[3 * r + 2] * inv[6 + c]
             for r in range(3) for c in range(3)]
        ax += w[0] * x + w[1] * y + w[2] * h
        ay += w[3] * x + w[4] * y + w[5] * h
    tx /= float(len(means))
    ty /= float(len(means))
    return np.array([ax, ay, math.atan2(ty, tx)])
