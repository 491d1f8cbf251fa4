"""GPU parity of the loop-candidate enumeration (kh_graph_*, through the C ABI): against the reference's own
MapperGraph::FindPossibleLoopClosure outputs (tests/golden/loop_candidates.npz) and, on a 6000-scan graph,
against the CPU oracle (oracle/loops.py).  Bar: identical chains, in order."""
import os

import numpy as np
import pytest

from slam_toolbox_amd import synth

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loop_candidates.npz"))


def test_chains_match_the_reference_for_every_scan_as_query(kartohip_lib):
    from slam_toolbox_amd.loop_search import MapperGraphSearch
    s = MapperGraphSearch()
    s.SetGraph(G["ref_xy"], G["adj_ptr"], G["adj_idx"])
    n = G["ref_xy"].shape[0]
    got = s.FindPossibleLoopClosures(np.arange(n), float(G["loop_search_maximum_distance"]),
                                     int(G["loop_match_minimum_chain_size"]))
    rows = [(q, a, b) for q in range(n) for a, b in got[q]]
    assert np.array_equal(np.asarray(rows, dtype=np.int32).reshape(-1, 3), G["chains"])
    s.close()


def _adjacency(n, edges):
    nbr = [[] for _ in range(n)]
    for a, b in edges:                     # Vertex::AddEdge appends to both ends in insertion order
        nbr[a].append(b)
        nbr[b].append(a)
    ptr = np.zeros(n + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([len(v) for v in nbr])
    return ptr, np.asarray([w for v in nbr for w in v], dtype=np.int32)


@pytest.mark.parametrize("max_distance,min_chain", [(3.0, 10), (6.0, 4), (0.4, 1)])
def test_large_graph_against_the_oracle(kartohip_lib, max_distance, min_chain):
    from oracle import loops
    from slam_toolbox_amd.loop_search import MapperGraphSearch
    g = synth.make_pose_graph(6000, 16000, seed=77)
    xy = g["truth"][:, :2].copy()
    # odometry chain + a tenth of the near-pair links: revisited aisles then hold unlinked runs (= loop candidates)
    edges = np.concatenate([g["edges"][:5999], g["edges"][5999::10]])
    ptr, idx = _adjacency(xy.shape[0], edges)
    s = MapperGraphSearch()
    s.SetGraph(xy, ptr, idx)
    queries = np.arange(0, xy.shape[0], 97, dtype=np.int32)
    got = s.FindPossibleLoopClosures(queries, max_distance, min_chain)
    n_chains = 0
    for k, q in enumerate(queries):
        ref = loops.find_possible_loop_closures(int(q), xy, ptr, idx, max_distance, min_chain)
        assert got[k] == ref, (q, got[k][:3], ref[:3])
        n_chains += len(ref)
    assert n_chains > 0
    # one query at a time is answered from the host copy of the store (a mapper's per-scan question): same chains as the
    # kernel gave for the batch
    for k in range(0, len(queries), 7):
        assert s.FindPossibleLoopClosures(queries[k:k + 1], max_distance, min_chain)[0] == got[k]
    # moved scans, same topology
    xy2 = xy + 0.05 * np.sin(np.arange(xy.size).reshape(xy.shape))
    s.SetPositions(xy2)
    got2 = s.FindPossibleLoopClosures(queries[:8], max_distance, min_chain)
    for k, q in enumerate(queries[:8]):
        assert got2[k] == loops.find_possible_loop_closures(int(q), xy2, ptr, idx, max_distance, min_chain)
    s.close()


def test_near_chains_closest_scan_and_weighted_mean_match_the_reference(kartohip_lib):
    """The neighbourhood-sized members of the row through the C ABI (host arithmetic inside the library):
    FindNearChains with its chain order, GetClosestScanToPose, ComputeWeightedMean -- against what the
    reference's own functions returned (tests/golden/loop_candidates.npz)."""
    import os
    from slam_toolbox_amd.loop_search import ComputeWeightedMean, MapperGraphSearch
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "loop_candidates.npz"))
    s = MapperGraphSearch()
    s.SetGraph(G["ref_xy"], G["adj_ptr"], G["adj_idx"])
    d = float(G["link_scan_maximum_distance"])
    rows = []
    for q in range(G["ref_xy"].shape[0]):
        for first, last in s.FindNearChains(q, d):
            rows.append((q, first, last, s.GetClosestScanToPose(np.arange(first, last + 1), G["ref_xy"][q])))
    assert np.array_equal(np.asarray(rows, dtype=np.int32).reshape(-1, 4), G["near_chains"])
    assert s.GetClosestScanToPose([], [0.0, 0.0]) == -1
    s.close()
    for k, row, want in zip(G["wm_k"], G["wm_in"], G["wm_out"]):
        k = int(k)
        covs = np.zeros((k, 9))
        covs[:, 0] = row[:k, 3]; covs[:, 4] = row[:k, 4]; covs[:, 8] = row[:k, 5]
        covs[:, 1] = row[:k, 6]; covs[:, 3] = row[:k, 6]
        assert np.array_equal(ComputeWeightedMean(row[:k, :3], covs), want)
