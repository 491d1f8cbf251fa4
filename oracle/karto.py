"""TEST INFRASTRUCTURE: ctypes binding of oracle/libkarto_oracle.so (plain-C restatement of the
karto scan matcher; see karto_oracle.c for the reference citations)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libkarto_oracle.so")

dptr = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
iptr = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class KoScan(C.Structure):
    _fields_ = [("n", C.c_int32), ("ranges", C.POINTER(C.c_double)), ("points", C.POINTER(C.c_double)),
                ("sensor_pose", C.c_double * 3)]


class KoPoseResponse(C.Structure):
    _fields_ = [("response", C.c_double), ("x", C.c_double), ("y", C.c_double), ("heading", C.c_double)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("karto_oracle.c", "occupancy_oracle.c")]
    if force or not os.path.exists(_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_PATH) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libkarto_oracle.so"])
    return _PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ko_scan_points.argtypes = [dptr, C.c_int32, dptr, C.c_double, C.c_double, dptr]
        L.ko_matcher_create.restype = C.c_void_p
        L.ko_matcher_create.argtypes = [C.c_double] * 4
        L.ko_matcher_destroy.argtypes = [C.c_void_p]
        L.ko_matcher_set_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                            C.c_double, C.c_double, C.c_double, C.c_double]
        L.ko_matcher_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.ko_find_valid_points.restype = C.c_int32
        L.ko_find_valid_points.argtypes = [C.POINTER(KoScan), dptr, dptr]
        L.ko_add_scans.argtypes = [C.c_void_p, C.POINTER(KoScan), C.c_int32, dptr]
        L.ko_center_grid.argtypes = [C.c_void_p, dptr]
        L.ko_compute_offsets.argtypes = [C.c_void_p, C.POINTER(KoScan), C.c_double, C.c_double, C.c_double]
        L.ko_get_response.restype = C.c_double
        L.ko_get_response.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
        L.ko_correlate_scan.restype = C.c_double
        L.ko_correlate_scan.argtypes = [C.c_void_p, C.POINTER(KoScan), dptr] + [C.c_double] * 6 + [C.c_int, dptr, dptr, C.c_int]
        L.ko_match_scan.restype = C.c_double
        L.ko_match_scan.argtypes = [C.c_void_p, C.POINTER(KoScan), C.POINTER(KoScan), C.c_int32, C.c_int, C.c_int, dptr, dptr]
        L.ko_grid_info.argtypes = [C.c_void_p, iptr, dptr]
        L.ko_grid_data.restype = C.POINTER(C.c_uint8)
        L.ko_grid_data.argtypes = [C.c_void_p]
        L.ko_kernel_data.restype = C.POINTER(C.c_uint8)
        L.ko_kernel_data.argtypes = [C.c_void_p]
        L.ko_lookup_data.restype = C.POINTER(C.c_int32)
        L.ko_lookup_data.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ko_probs_data.restype = C.POINTER(C.c_double)
        L.ko_probs_data.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ko_volume.restype = C.POINTER(KoPoseResponse)
        L.ko_volume.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ko_world_to_grid_index.restype = C.c_int32
        L.ko_world_to_grid_index.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def scan_points(ranges, sensor_pose, laser):
    ranges = _d(ranges)
    out = np.zeros((ranges.shape[0], 2))
    lib().ko_scan_points(ranges, ranges.shape[0], _d(sensor_pose), laser.min_angle, laser.ang_res, out)
    return out


class Scan:
    """ranges + unfiltered world points + sensor pose (what the matcher reads from a LocalizedRangeScan)."""

    def __init__(self, ranges, sensor_pose, laser=None, points=None):
        self.ranges = _d(ranges)
        self.sensor_pose = _d(sensor_pose).copy()
        if points is None:
            points = scan_points(self.ranges, self.sensor_pose, laser)
        self.points = _d(points)
        self.n = self.ranges.shape[0]

    def c(self) -> KoScan:
        s = KoScan()
        s.n = self.n
        s.ranges = self.ranges.ctypes.data_as(C.POINTER(C.c_double))
        s.points = self.points.ctypes.data_as(C.POINTER(C.c_double))
        for i in range(3):
            s.sensor_pose[i] = self.sensor_pose[i]
        return s

    def with_sensor_pose(self, pose, laser):
        """SetSensorPose + Update (Karto.h:5552-5557): points are recomputed from the new pose."""
        return Scan(self.ranges, pose, laser)


def _arr(scans):
    a = (KoScan * max(1, len(scans)))()
    for i, s in enumerate(scans):
        a[i] = s.c()
    return a


class Matcher:
    def __init__(self, search_size, resolution, smear, range_threshold, params=None, threads=1):
        self.h = lib().ko_matcher_create(search_size, resolution, smear, range_threshold)
        if not self.h:
            raise ValueError("ko_matcher_create: invalid parameters")
        if params is not None:
            self.set_params(**params)
        lib().ko_matcher_set_threads(self.h, threads)

    def set_params(self, coarse_search_angle_offset, coarse_angle_resolution, fine_search_angle_offset,
                   use_response_expansion, distance_variance_penalty, minimum_distance_penalty,
                   angle_variance_penalty, minimum_angle_penalty):
        """Same convention as the reference setters: the two variance penalties are squared on the way in
        (Mapper.cpp:2562-2570)."""
        lib().ko_matcher_set_params(self.h, coarse_search_angle_offset, coarse_angle_resolution,
                                    fine_search_angle_offset, int(use_response_expansion),
                                    distance_variance_penalty * distance_variance_penalty, minimum_distance_penalty,
                                    angle_variance_penalty * angle_variance_penalty, minimum_angle_penalty)

    def match_scan(self, scan, base, do_penalize=True, do_refine=True):
        mean = np.zeros(3)
        cov = np.zeros(9)
        cs = scan.c()
        r = lib().ko_match_scan(self.h, C.byref(cs), _arr(base), len(base), int(do_penalize), int(do_refine), mean, cov)
        return r, mean, cov.reshape(3, 3)

    def add_scans(self, scan, base):
        lib().ko_center_grid(self.h, _d(scan.sensor_pose))
        lib().ko_add_scans(self.h, _arr(base), len(base), _d(scan.sensor_pose[:2]))

    def correlate_scan(self, scan, center, off, res, ang_off, ang_res, do_penalize, fine, cov_in=None):
        mean = np.zeros(3)
        cov = np.zeros(9) if cov_in is None else _d(cov_in).reshape(9).copy()
        cs = scan.c()
        r = lib().ko_correlate_scan(self.h, C.byref(cs), _d(center), off[0], off[1], res[0], res[1], ang_off, ang_res,
                                    int(do_penalize), mean, cov, int(fine))
        return r, mean, cov.reshape(3, 3)

    def grid_info(self):
        i = np.zeros(9, dtype=np.int32)
        d = np.zeros(3)
        lib().ko_grid_info(self.h, i, d)
        keys = ["width", "height", "width_step", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size", "data_size"]
        out = dict(zip(keys, (int(v) for v in i)))
        out.update(offset_x=d[0], offset_y=d[1], scale=d[2])
        return out

    def grid(self):
        n = self.grid_info()["data_size"]
        return np.ctypeslib.as_array(lib().ko_grid_data(self.h), shape=(n,)).copy()

    def kernel(self):
        k = self.grid_info()["kernel_size"]
        return np.ctypeslib.as_array(lib().ko_kernel_data(self.h), shape=(k * k,)).copy().reshape(k, k)

    def compute_offsets(self, scan, angle_center, ang_off, ang_res):
        cs = scan.c()
        lib().ko_compute_offsets(self.h, C.byref(cs), angle_center, ang_off, ang_res)

    def lookup_table(self):
        na, npnt = C.c_int32(), C.c_int32()
        p = lib().ko_lookup_data(self.h, C.byref(na), C.byref(npnt))
        return np.ctypeslib.as_array(p, shape=(na.value * npnt.value,)).copy().reshape(na.value, npnt.value)

    def get_response(self, angle_index, grid_index):
        return lib().ko_get_response(self.h, angle_index, grid_index)

    def world_to_grid_index(self, x, y):
        return lib().ko_world_to_grid_index(self.h, x, y)

    def probs(self):
        side, ws = C.c_int32(), C.c_int32()
        p = lib().ko_probs_data(self.h, C.byref(side), C.byref(ws))
        return np.ctypeslib.as_array(p, shape=(side.value * ws.value,)).copy().reshape(side.value, ws.value)[:, :side.value]

    def volume(self):
        """(ny, nx, na, 4) array of (response, x, y, heading) from the last CorrelateScan."""
        nx, ny, na = C.c_int32(), C.c_int32(), C.c_int32()
        p = lib().ko_volume(self.h, C.byref(nx), C.byref(ny), C.byref(na))
        n = nx.value * ny.value * na.value
        buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n * 4,)).copy()
        return buf.reshape(ny.value, nx.value, na.value, 4)

    def find_valid_points(self, scan, viewpoint):
        out = np.zeros((scan.n, 2))
        cs = scan.c()
        n = lib().ko_find_valid_points(C.byref(cs), _d(viewpoint), out)
        return out[:n]

    def __del__(self):
        try:
            lib().ko_matcher_destroy(self.h)
        except Exception:
            pass


def occupancy_from_scans(width, height, offset, resolution, scans, laser, min_pass_through=2, occupancy_threshold=0.1):
    """OccupancyGrid::CreateFromScans (occupancy_oracle.c): returns (cells, pass_counts, hit_counts), each
    (height, width_step) with width_step = align8(width)."""
    L = lib()
    ws = (int(width) + 7) & ~7
    size = ws * int(height)
    passes = np.zeros(size, dtype=np.uint32)
    hits = np.zeros(size, dtype=np.uint32)
    cells = np.zeros(size, dtype=np.uint8)
    fn = L.ko_occupancy_from_scans
    fn.restype = None
    fn.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.POINTER(KoScan), C.c_double,
                   C.c_double, C.c_double, C.c_uint32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    arr = _arr(scans)
    fn(int(width), int(height), float(offset[0]), float(offset[1]), float(resolution), len(scans), arr,
       float(laser.range_threshold), float(laser.min_range), float(laser.max_range), int(min_pass_through),
       float(occupancy_threshold), passes.ctypes.data, hits.ctypes.data, cells.ctypes.data)
    return cells.reshape(height, ws), passes.reshape(height, ws), hits.reshape(height, ws)
