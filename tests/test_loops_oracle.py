"""CPU: the loop-candidate enumeration oracle (oracle/loops.py) against what the reference's own
MapperGraph::FindNearLinkedScans / FindPossibleLoopClosure returned (tests/golden/loop_candidates.npz)."""
import os

import numpy as np

from oracle import loops

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loop_candidates.npz"))


def test_near_linked_scans_match_the_reference_in_bfs_order():
    n = G["ref_xy"].shape[0]
    d = float(G["loop_search_maximum_distance"])
    for q in range(n):
        got = loops.near_linked_scans(q, G["ref_xy"], G["adj_ptr"], G["adj_idx"], d)
        ref = G["link_idx"][G["link_ptr"][q]: G["link_ptr"][q + 1]].tolist()
        assert got == ref, q


def test_loop_chains_match_the_reference():
    n = G["ref_xy"].shape[0]
    d = float(G["loop_search_maximum_distance"])
    k = int(G["loop_match_minimum_chain_size"])
    rows = []
    for q in range(n):
        for first, last in loops.find_possible_loop_closures(q, G["ref_xy"], G["adj_ptr"], G["adj_idx"], d, k):
            rows.append((q, first, last))
    assert len(G["chains"]) > 20
    assert np.array_equal(np.asarray(rows, dtype=np.int32).reshape(-1, 3), G["chains"])
