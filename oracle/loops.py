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
