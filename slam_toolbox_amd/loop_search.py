"""Python mirror of the candidate-enumeration part of karto::MapperGraph over the C ABI
(kh_graph_*): FindNearLinkedScans + FindPossibleLoopClosure for a batch of query scans on the GPU.  Same
argument meaning as the reference: positions are GetReferencePose(use_scan_barycenter) of the scans in
scan-list order, adjacency in Vertex::GetAdjacentVertices order."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class MapperGraphSearch:
    def __init__(self, device: int = 0):
        h = C.c_void_p()
        capi.check(capi.lib().kh_graph_create(device, C.byref(h)), "kh_graph_create")
        self._h = h
        self.n = 0

    def SetGraph(self, ref_xy, adj_ptr, adj_idx):
        ref_xy = np.ascontiguousarray(ref_xy, dtype=np.float64).reshape(-1, 2)
        adj_ptr = np.ascontiguousarray(adj_ptr, dtype=np.int32)
        adj_idx = np.ascontiguousarray(adj_idx, dtype=np.int32)
        if adj_idx.size == 0:
            adj_idx = np.zeros(1, dtype=np.int32)
        capi.check(capi.lib().kh_graph_set(self._h, ref_xy.shape[0], ref_xy.reshape(-1), adj_ptr, adj_idx), "kh_graph_set")
        self.n = ref_xy.shape[0]

    def SetPositions(self, ref_xy):
        """After CorrectPoses(): same topology, moved scans."""
        ref_xy = np.ascontiguousarray(ref_xy, dtype=np.float64).reshape(-1, 2)
        capi.check(capi.lib().kh_graph_set_positions(self._h, ref_xy.shape[0], ref_xy.reshape(-1)), "kh_graph_set_positions")

    def FindPossibleLoopClosures(self, query_scans, loop_search_maximum_distance, loop_match_minimum_chain_size):
        """-> list (one entry per query) of [(first, last), ...]: every chain successive
        MapperGraph::FindPossibleLoopClosure calls would return for that scan (Mapper.cpp:1960-2010)."""
        q = np.ascontiguousarray(query_scans, dtype=np.int32)
        begin = np.zeros(q.size + 1, dtype=np.int32)
        cap = max(16, 4 * q.size)
        total = C.c_int32(0)
        while True:
            chains = np.zeros(2 * cap, dtype=np.int32)
            capi.check(capi.lib().kh_graph_find_loop_candidates(self._h, q.size, q, float(loop_search_maximum_distance),
                                                                int(loop_match_minimum_chain_size), begin, chains, cap,
                                                                C.byref(total)), "kh_graph_find_loop_candidates")
            if total.value <= cap:
                break
            cap = total.value
        chains = chains.reshape(-1, 2)
        return [[(int(a), int(b)) for a, b in chains[begin[i]: begin[i + 1]]] for i in range(q.size)]

    def FindNearChains(self, query_scan, link_scan_maximum_distance):
        """MapperGraph::FindNearChains (Mapper.cpp:1683-1793) -> [(first, last), ...] in the reference's order."""
        cap = 64
        total = C.c_int32(0)
        while True:
            chains = np.zeros(2 * cap, dtype=np.int32)
            capi.check(capi.lib().kh_graph_find_near_chains(self._h, int(query_scan), float(link_scan_maximum_distance),
                                                            chains, cap, C.byref(total)), "kh_graph_find_near_chains")
            if total.value <= cap:
                break
            cap = total.value
        return [(int(a), int(b)) for a, b in chains.reshape(-1, 2)[:total.value]]

    def GetClosestScanToPose(self, scans, pose_xy):
        """MapperGraph::GetClosestScanToPose (Mapper.cpp:1563-1582); -1 for an empty list."""
        scans = np.ascontiguousarray(scans, dtype=np.int32)
        out = C.c_int32(-1)
        capi.check(capi.lib().kh_graph_closest_scan_to_pose(self._h, scans if scans.size else np.zeros(1, dtype=np.int32),
                                                            scans.size, np.ascontiguousarray(pose_xy, dtype=np.float64)[:2].copy(),
                                                            C.byref(out)), "kh_graph_closest_scan_to_pose")
        return out.value

    def last_kernel_ms(self):
        return capi.lib().kh_graph_last_kernel_ms(self._h)

    def close(self):
        if self._h:
            capi.lib().kh_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ComputeWeightedMean(means, covariances):
    """MapperGraph::ComputeWeightedMean (Mapper.cpp:1914-1958)."""
    means = np.ascontiguousarray(means, dtype=np.float64).reshape(-1, 3)
    covs = np.ascontiguousarray(covariances, dtype=np.float64).reshape(-1, 9)
    out = np.zeros(3)
    capi.check(capi.lib().kh_weighted_mean(means.shape[0], means.reshape(-1), covs.reshape(-1), out), "kh_weighted_mean")
    return out
