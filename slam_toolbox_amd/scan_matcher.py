"""Host-side mirror of the reference's scan-matcher interface (karto::ScanMatcher,
lib/karto_sdk/include/karto_sdk/Mapper.h:1322-1544) over the C ABI of libkartohip.so.

Same names, argument meaning and error behaviour as the reference so that the parity tests read
like tests of the reference class:

    matcher = ScanMatcher.Create(mapper_params, searchSize, resolution, smearDeviation, rangeThreshold)
    response, mean, cov = matcher.MatchScan(scan, base_scans, doPenalize=True, doRefineMatch=True)
    response, mean, cov = matcher.CorrelateScan(scan, searchCenter, searchSpaceOffset, searchSpaceResolution,
                                                searchAngleOffset, searchAngleResolution, doPenalize,
                                                covariance, doingFineMatch)

All scoring runs on the GPU through the library; nothing here computes a response."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class MapperParams:
    """The eight karto::Mapper parameters ScanMatcher reads (Mapper.cpp:590-627, 671-682).  As with the
    reference setters (Mapper.cpp:2562-2570) distance/angle variance penalties are given UNSQUARED and
    squared on the way in.  Defaults = Mapper::InitializeParameters (Mapper.cpp:2250-2293)."""
    coarse_search_angle_offset: float = 20 * 0.01745329251994329577
    coarse_angle_resolution: float = 2 * 0.01745329251994329577
    fine_search_angle_offset: float = 0.2 * 0.01745329251994329577
    use_response_expansion: bool = False
    distance_variance_penalty: float = 0.3
    minimum_distance_penalty: float = 0.5
    angle_variance_penalty: float = 20 * 0.01745329251994329577
    minimum_angle_penalty: float = 0.9

    def c(self) -> capi.KhMatchParams:
        p = capi.KhMatchParams()
        p.coarse_search_angle_offset = self.coarse_search_angle_offset
        p.coarse_angle_resolution = self.coarse_angle_resolution
        p.fine_search_angle_offset = self.fine_search_angle_offset
        p.use_response_expansion = int(self.use_response_expansion)
        p.distance_variance_penalty = self.distance_variance_penalty * self.distance_variance_penalty
        p.minimum_distance_penalty = self.minimum_distance_penalty
        p.angle_variance_penalty = self.angle_variance_penalty * self.angle_variance_penalty
        p.minimum_angle_penalty = self.minimum_angle_penalty
        return p


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class LocalizedRangeScan:
    """What the matcher reads from karto::LocalizedRangeScan (Karto.h:5380-5760): range readings, sensor
    pose and the unfiltered point readings computed by Update() (Karto.h:5644-5704)."""

    def __init__(self, ranges, sensor_pose, min_angle, angular_resolution):
        self.ranges = _d(ranges)
        self.min_angle = float(min_angle)
        self.angular_resolution = float(angular_resolution)
        self.SetSensorPose(sensor_pose)

    _resident = None
    _c_points = None
    _c_ranges = None
    _c = None          # the kh_scan view of this scan, rebuilt after Update() / MakeResident (building one costs more than a match's launch)

    def SetSensorPose(self, pose):
        self._c = None
        self.sensor_pose = _d(pose).copy()
        self.points = np.zeros((self.ranges.shape[0], 2))
        capi.check(capi.lib().kh_scan_points(self.ranges, self.ranges.shape[0], self.sensor_pose, self.min_angle,
                                             self.angular_resolution, self.points), "kh_scan_points")
        if self._resident is not None:
            self._resident.upload(self.points)           # the device copy follows Update() (Karto.h:5644-5704)

    def MakeResident(self, device=0):
        """Keeps the point readings in HBM as well (kh_scan.device_points_xy): as a BASE scan of AddScans / MatchScan the
        scan is then read where it lies instead of being uploaded with every call."""
        from .comm import DeviceBuffer
        if self._resident is None and self.points.size > 0:
            self._resident = DeviceBuffer(self.points.size, device)
            self._resident.upload(self.points)
            self._c = None
        return self

    def GetSensorPose(self):
        return self.sensor_pose.copy()

    def GetNumberOfRangeReadings(self):
        return self.ranges.shape[0]

    def c(self) -> capi.KhScan:
        # (the struct holds raw pointers into ranges / points and the pose by value: it is rebuilt when any of them was replaced
        # or the pose edited in place)
        if (self._c is not None and self._c_points is self.points and self._c_ranges is self.ranges
                and all(self._c.sensor_pose[i] == self.sensor_pose[i] for i in range(3))):
            return self._c
        s = capi.KhScan()
        s.n = self.ranges.shape[0]
        s.ranges = self.ranges.ctypes.data_as(C.POINTER(C.c_double))
        s.points_xy = self.points.ctypes.data_as(C.POINTER(C.c_double))
        for i in range(3):
            s.sensor_pose[i] = self.sensor_pose[i]
        s.device_points_xy = self._resident.ptr if self._resident is not None else None
        self._c, self._c_points, self._c_ranges = s, self.points, self.ranges
        return s


def _scan_array(scans):
    arr = (capi.KhScan * max(1, len(scans)))()
    for i, s in enumerate(scans):
        arr[i] = s.c()
    return arr


class ScanMatcher:
    def __init__(self, handle, max_batch):
        self._h = handle
        self.max_batch = max_batch

    @staticmethod
    def Create(mapper_params: MapperParams, searchSize, resolution, smearDeviation, rangeThreshold,
               device: int = 0, max_batch: int = 1):
        """ScanMatcher::Create (Mapper.cpp:477-522): returns None for invalid parameters like the reference
        returns NULL; raises when no GPU is available (no CPU fallback)."""
        h = C.c_void_p()
        rc = capi.lib().kh_matcher_create(searchSize, resolution, smearDeviation, rangeThreshold, device, max_batch,
                                          C.byref(h))
        if rc == capi.KH_ERR_INVALID_ARG:
            return None
        capi.check(rc, "kh_matcher_create")
        m = ScanMatcher(h, max_batch)
        m.SetParams(mapper_params)
        return m

    def SetParams(self, mapper_params: MapperParams):
        p = mapper_params.c()
        capi.check(capi.lib().kh_matcher_set_params(self._h, C.byref(p)), "kh_matcher_set_params")

    def MatchScan(self, scan, base_scans, doPenalize=True, doRefineMatch=True):
        mean = np.zeros(3)
        cov = np.zeros(9)
        resp = C.c_double(0.0)
        cs = scan.c()
        rc = capi.lib().kh_matcher_match(self._h, C.byref(cs), _scan_array(base_scans), len(base_scans),
                                         int(doPenalize), int(doRefineMatch), mean, cov, C.byref(resp))
        if rc == capi.KH_ERR_SEARCH:
            raise RuntimeError("Mapper FATAL ERROR - Unable to find best position")   # Mapper.cpp:786-796, 828
        capi.check(rc, "kh_matcher_match")
        return resp.value, mean, cov.reshape(3, 3)

    @staticmethod
    def pack_batch(scans, base_lists):
        """ctypes marshalling of a MatchScanBatch input, reusable across calls (a C++ caller hands the
        kh_scan arrays over directly; in Python building ~25 structs per pair costs more than the match)."""
        flat = [b for lst in base_lists for b in lst]
        begin = np.zeros(len(scans) + 1, dtype=np.int32)
        begin[1:] = np.cumsum([len(lst) for lst in base_lists])
        return (_scan_array(scans), _scan_array(flat), begin, len(scans), (scans, flat))

    def MatchScanBatch(self, scans, base_lists, doPenalize=True, doRefineMatch=True, packed=None):
        if packed is not None:
            q_arr, b_arr, begin, n, _keep = packed
            means = np.zeros(3 * n)
            covs = np.zeros(9 * n)
            resp = np.zeros(n)
            status = np.zeros(n, dtype=np.int32)
            capi.check(capi.lib().kh_matcher_match_batch(self._h, n, q_arr, b_arr, begin, int(doPenalize),
                                                         int(doRefineMatch), means, covs, resp, status),
                       "kh_matcher_match_batch")
            return resp, means.reshape(n, 3), covs.reshape(n, 3, 3), status
        n = len(scans)
        flat = [b for lst in base_lists for b in lst]
        begin = np.zeros(n + 1, dtype=np.int32)
        begin[1:] = np.cumsum([len(lst) for lst in base_lists])
        means = np.zeros(3 * n)
        covs = np.zeros(9 * n)
        resp = np.zeros(n)
        status = np.zeros(n, dtype=np.int32)
        capi.check(capi.lib().kh_matcher_match_batch(self._h, n, _scan_array(scans), _scan_array(flat), begin,
                                                     int(doPenalize), int(doRefineMatch), means, covs, resp, status),
                   "kh_matcher_match_batch")
        return resp, means.reshape(n, 3), covs.reshape(n, 3, 3), status

    def AddScans(self, scan, base_scans, slot=0):
        cs = scan.c()
        capi.check(capi.lib().kh_matcher_add_scans(self._h, slot, C.byref(cs), _scan_array(base_scans), len(base_scans)),
                   "kh_matcher_add_scans")

    def CorrelateScan(self, scan, searchCenter, searchSpaceOffset, searchSpaceResolution, searchAngleOffset,
                      searchAngleResolution, doPenalize, covariance=None, doingFineMatch=False, slot=0):
        mean = np.zeros(3)
        cov = np.zeros(9) if covariance is None else _d(covariance).reshape(9).copy()
        resp = C.c_double(0.0)
        cs = scan.c()
        rc = capi.lib().kh_matcher_correlate(self._h, slot, C.byref(cs), _d(searchCenter), _d(searchSpaceOffset),
                                             _d(searchSpaceResolution), searchAngleOffset, searchAngleResolution,
                                             int(doPenalize), int(doingFineMatch), mean, cov, C.byref(resp))
        if rc == capi.KH_ERR_SEARCH:
            raise RuntimeError("Mapper FATAL ERROR - Unable to find best position")
        capi.check(rc, "kh_matcher_correlate")
        return resp.value, mean, cov.reshape(3, 3)

    def CorrelateScanBatch(self, scans, centers, searchSpaceOffset, searchSpaceResolution, searchAngleOffset,
                           searchAngleResolution, doPenalize, doingFineMatch=False, scan_array=None):
        n = len(scans) if scan_array is None else scan_array[1]
        arr = _scan_array(scans) if scan_array is None else scan_array[0]
        means = np.zeros(3 * n)
        covs = np.zeros(9 * n)
        resp = np.zeros(n)
        status = np.zeros(n, dtype=np.int32)
        capi.check(capi.lib().kh_matcher_correlate_batch(self._h, n, arr, _d(centers).reshape(-1),
                                                         _d(searchSpaceOffset), _d(searchSpaceResolution),
                                                         searchAngleOffset, searchAngleResolution, int(doPenalize),
                                                         int(doingFineMatch), means, covs, resp, status),
                   "kh_matcher_correlate_batch")
        return resp, means.reshape(n, 3), covs.reshape(n, 3, 3), status

    def ComputePositionalCovariance(self, bestPose, bestResponse, searchCenter, searchSpaceOffset, searchSpaceResolution,
                                    searchAngleResolution, slot=0):
        """ScanMatcher::ComputePositionalCovariance (Mapper.cpp:874-966) on the probabilities of the last coarse search"""
        cov = np.zeros(9)
        capi.check(capi.lib().kh_matcher_positional_covariance(self._h, slot, _d(bestPose), float(bestResponse), _d(searchCenter),
                                                               _d(searchSpaceOffset), _d(searchSpaceResolution),
                                                               float(searchAngleResolution), cov), "kh_matcher_positional_covariance")
        return cov.reshape(3, 3)

    def ComputeAngularCovariance(self, scan, bestPose, bestResponse, searchCenter, searchAngleOffset, searchAngleResolution, slot=0):
        """ScanMatcher::ComputeAngularCovariance (Mapper.cpp:977-1025): returns the theta-theta variance"""
        cov = np.zeros(9)
        cs = scan.c()
        capi.check(capi.lib().kh_matcher_angular_covariance(self._h, slot, C.byref(cs), _d(bestPose), float(bestResponse),
                                                            _d(searchCenter), float(searchAngleOffset), float(searchAngleResolution),
                                                            cov), "kh_matcher_angular_covariance")
        return cov[8]

    # ---- introspection (parity tests / bench) ----
    def grid_info(self, slot=0):
        g = capi.KhGridInfo()
        capi.check(capi.lib().kh_matcher_grid_info(self._h, slot, C.byref(g)), "kh_matcher_grid_info")
        return {k: getattr(g, k) for k, _ in capi.KhGridInfo._fields_}

    def GetCorrelationGrid(self, slot=0):
        n = self.grid_info(slot)["data_size"]
        out = np.zeros(n, dtype=np.uint8)
        capi.check(capi.lib().kh_matcher_read_grid(self._h, slot, out), "kh_matcher_read_grid")
        return out

    def kernel(self):
        k = self.grid_info()["kernel_size"]
        out = np.zeros(k * k, dtype=np.uint8)
        capi.check(capi.lib().kh_matcher_read_kernel(self._h, out), "kh_matcher_read_kernel")
        return out.reshape(k, k)

    def lookup_table(self, slot=0):
        na, npnt = C.c_int32(), C.c_int32()
        capi.check(capi.lib().kh_matcher_read_lookup(self._h, slot, C.byref(na), C.byref(npnt), None), "read_lookup")
        out = np.zeros((na.value, npnt.value), dtype=np.int32)
        capi.check(capi.lib().kh_matcher_read_lookup(self._h, slot, C.byref(na), C.byref(npnt),
                                                     out.ctypes.data_as(C.c_void_p)), "read_lookup")
        return out

    def set_debug(self, keep_response_volume: bool, lds_score: bool = False, dense_score: bool = False,
                  force_chunks: bool = False, no_dual_copy: bool = False, mfma_score: bool = False,
                  windowed_score: bool = False, no_fused_match: bool = False):
        """lds_score: the LDS-staged scoring kernels for every search they can take (default: the large ones only);
        windowed_score: for none (the windowed kernel scores everything); no_fused_match: MatchScan takes the general
        (batch) path instead of the fused path of one match."""
        capi.check(capi.lib().kh_matcher_set_debug(self._h, int(bool(keep_response_volume)) | (2 if lds_score else 0) |
                                                   (4 if dense_score else 0) | (8 if force_chunks else 0) |
                                                   (16 if no_dual_copy else 0) | (32 if mfma_score else 0) |
                                                   (64 if windowed_score else 0) | (128 if no_fused_match else 0)),
                   "kh_matcher_set_debug")

    def volume(self, slot=0, responses=True):
        nx, ny, na = C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(capi.lib().kh_matcher_read_volume(self._h, slot, C.byref(nx), C.byref(ny), C.byref(na), None, None),
                   "read_volume")
        sums = np.zeros((ny.value, nx.value, na.value), dtype=np.int32)
        resp = np.zeros((ny.value, nx.value, na.value)) if responses else None
        capi.check(capi.lib().kh_matcher_read_volume(self._h, slot, C.byref(nx), C.byref(ny), C.byref(na),
                                                     sums.ctypes.data_as(C.c_void_p),
                                                     resp.ctypes.data_as(C.c_void_p) if responses else None),
                   "read_volume")
        return sums, resp

    def seq_stats(self):
        """counters of the fused path of ONE MatchScan (kh_matcher_seq_stats)"""
        out = (C.c_int64 * 8)()
        capi.check(capi.lib().kh_matcher_seq_stats(self._h, out), "kh_matcher_seq_stats")
        keys = ("calls", "fine_on_device", "fine_fallbacks", "fine_mismatches", "coarse_fallbacks", "fused_score", "ineligible", "ineligible_reason")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def stream(self):
        return capi.lib().kh_matcher_stream(self._h)

    def profile(self, enable=True):
        sm, rm = C.c_double(), C.c_double()
        sl, rl = C.c_int64(), C.c_int64()
        capi.check(capi.lib().kh_matcher_profile(self._h, int(enable), C.byref(sm), C.byref(sl), C.byref(rm), C.byref(rl)),
                   "kh_matcher_profile")
        return {"score_ms": sm.value, "score_launches": sl.value, "raster_ms": rm.value, "raster_launches": rl.value}

    def profile_side(self):
        """GPU ms of the table / list kernel (K2) and the tie kernel (K4) of the launches profiled since the last call"""
        o, t = C.c_double(), C.c_double()
        capi.check(capi.lib().kh_matcher_profile_side(self._h, C.byref(o), C.byref(t)), "kh_matcher_profile_side")
        return {"offsets_ms": o.value, "ties_ms": t.value}

    def score_loads(self, reset=True):
        """wave-level dword-load instructions (256 B each) K3 issued for the searches run while profiling was on"""
        n = C.c_int64()
        capi.check(capi.lib().kh_matcher_score_loads(self._h, C.byref(n), int(reset)), "kh_matcher_score_loads")
        return n.value

    def close(self):
        if self._h:
            capi.lib().kh_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def LoopClosureBatch(coarse: "ScanMatcher", fine: "ScanMatcher", scans, base_lists, min_angle: float, angular_resolution: float,
                     minimum_response_coarse: float, maximum_variance_coarse: float, pieces: int = 1, packed=None):
    """MapperGraph::TryCloseLoop's coarse match, gate and fine match of the temporary scan at the coarse pose
    (Mapper.cpp:1515-1549) for a batch of candidate chains: kh_loop_closure_batch.  Returns a dict of arrays; the fine_*
    rows of chains that did not pass are NaN."""
    q_arr, b_arr, begin, n, _keep = packed if packed is not None else ScanMatcher.pack_batch(scans, base_lists)
    cm, cc, cr = np.zeros(3 * n), np.zeros(9 * n), np.zeros(n)
    fm, fc, fr = np.full(3 * n, np.nan), np.full(9 * n, np.nan), np.full(n, np.nan)
    passed = np.zeros(n, dtype=np.int32)
    capi.check(capi.lib().kh_loop_closure_batch(coarse._h, fine._h, n, q_arr, b_arr, begin, min_angle, angular_resolution,
                                                minimum_response_coarse, maximum_variance_coarse, pieces, cm, cc, cr, passed, fm, fc, fr),
               "kh_loop_closure_batch")
    return {"coarse_response": cr, "coarse_mean": cm.reshape(n, 3), "coarse_covariance": cc.reshape(n, 3, 3), "passed": passed.astype(bool),
            "fine_response": fr, "fine_mean": fm.reshape(n, 3), "fine_covariance": fc.reshape(n, 3, 3)}


class ScanMatcherGroup:
    """kh_matcher_group: one ScanMatcher per entry of `devices` in ONE process; MatchScanBatch deals candidate i to member
    i % len(devices) (a host thread per member) and returns the results in candidate order -- the in-process form of the
    multi-GPU loop-closure batch (MapperGraph::TryCloseLoop's candidates, Mapper.cpp:1500-1561).  The same device may be
    listed more than once."""

    def __init__(self, mapper_params: MapperParams, searchSize, resolution, smearDeviation, rangeThreshold, devices,
                 max_batch_per_member: int = 64):
        self.devices = np.asarray(list(devices), dtype=np.int32)
        h = C.c_void_p()
        capi.check(capi.lib().kh_matcher_group_create(searchSize, resolution, smearDeviation, rangeThreshold, self.devices,
                                                      len(self.devices), max_batch_per_member, C.byref(h)), "kh_matcher_group_create")
        self._h = h
        p = mapper_params.c()
        capi.check(capi.lib().kh_matcher_group_set_params(self._h, C.byref(p)), "kh_matcher_group_set_params")

    def __len__(self):
        return int(capi.lib().kh_matcher_group_size(self._h))

    def pack_batch(self, scans, base_lists, device_points=None):
        """ctypes marshalling of a batch, reusable across calls (see ScanMatcher.pack_batch)"""
        q_arr, b_arr, begin, n, keep = ScanMatcher.pack_batch(scans, base_lists)
        table = None
        if device_points is not None:
            table = np.ascontiguousarray(device_points, dtype=np.uint64)
            assert table.shape == (len(keep[1]), len(self))
        return (q_arr, b_arr, begin, n, keep, table)

    def MatchScanBatch(self, scans, base_lists, doPenalize=True, doRefineMatch=True, device_points=None, packed=None):
        """device_points: None, or an (n_base_total, n_members) uint64 array of device addresses (0 = not resident)"""
        if packed is not None:
            q_arr, b_arr, begin, n, _keep, table = packed
            means, covs, resp, status = np.zeros(3 * n), np.zeros(9 * n), np.zeros(n), np.zeros(n, dtype=np.int32)
            capi.check(capi.lib().kh_matcher_group_match_batch(
                self._h, n, q_arr, b_arr, begin, None if table is None else table.ctypes.data_as(C.c_void_p),
                int(doPenalize), int(doRefineMatch), means, covs, resp, status), "kh_matcher_group_match_batch")
            return resp, means.reshape(n, 3), covs.reshape(n, 3, 3), status
        n = len(scans)
        flat = [b for lst in base_lists for b in lst]
        begin = np.zeros(n + 1, dtype=np.int32)
        begin[1:] = np.cumsum([len(lst) for lst in base_lists])
        means = np.zeros(3 * n)
        covs = np.zeros(9 * n)
        resp = np.zeros(n)
        status = np.zeros(n, dtype=np.int32)
        table = None
        if device_points is not None:
            table = np.ascontiguousarray(device_points, dtype=np.uint64)
            assert table.shape == (len(flat), len(self))
        capi.check(capi.lib().kh_matcher_group_match_batch(
            self._h, n, _scan_array(scans), _scan_array(flat), begin, None if table is None else table.ctypes.data_as(C.c_void_p),
            int(doPenalize), int(doRefineMatch), means, covs, resp, status), "kh_matcher_group_match_batch")
        return resp, means.reshape(n, 3), covs.reshape(n, 3, 3), status

    def close(self):
        if self._h:
            capi.lib().kh_matcher_group_destroy(self._h)
            self._h = None
