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


def _wm_cases():
    for k, row, out in zip(G["wm_k"], G["wm_in"], G["wm_out"]):
        k = int(k)
        means = row[:k, :3]
        covs = np.zeros((k, 9))
        covs[:, 0] = row[:k, 3]; covs[:, 4] = row[:k, 4]; covs[:, 8] = row[:k, 5]
        covs[:, 1] = row[:k, 6]; covs[:, 3] = row[:k, 6]
        yield means, covs, out


def test_near_chains_and_closest_scan_match_the_reference():
    """MapperGraph::FindNearChains (Mapper.cpp:1683-1793) for every scan of the reference graph, chain order
    included, and what GetClosestScanToPose (Mapper.cpp:1563-1582) picked inside every chain."""
    n = G["ref_xy"].shape[0]
    d = float(G["link_scan_maximum_distance"])
    rows = []
    for q in range(n):
        for first, last in loops.find_near_chains(q, G["ref_xy"], G["adj_ptr"], G["adj_idx"], d):
            rows.append((q, first, last, loops.closest_scan_to_pose(range(first, last + 1), G["ref_xy"], G["ref_xy"][q])))
    assert len(G["near_chains"]) > 100
    assert np.array_equal(np.asarray(rows, dtype=np.int32).reshape(-1, 4), G["near_chains"])


def test_weighted_mean_matches_the_reference_bit_for_bit():
    """MapperGraph::ComputeWeightedMean (Mapper.cpp:1914-1958) known answers printed by the reference build."""
    n = 0
    for means, covs, want in _wm_cases():
        got = loops.compute_weighted_mean(means, covs)
        assert np.array_equal(got, want), (got, want)
        n += 1
    assert n == 12
