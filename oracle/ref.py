"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libkarto_ref.so (the reference's own
karto_sdk sources compiled in place, see oracle/Makefile + oracle/ref_driver.cpp).

Used to pin the C restatement (oracle/karto_oracle.c) and to generate tests/golden/.  Exists
only where the .so has been built (the dev container; the built .so also travels to the GPU
box).  Never imported by the product package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libkarto_ref.so")

dptr = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
iptr = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def available() -> bool:
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_PATH)
        L.ref_init_laser.restype = C.c_int
        L.ref_init_laser.argtypes = [C.c_double] * 6
        L.ref_set_threads.argtypes = [C.c_int]
        L.ref_mapper_create.restype = C.c_void_p
        L.ref_mapper_destroy.argtypes = [C.c_void_p]
        L.ref_mapper_set_match_params.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int,
                                                  C.c_double, C.c_double, C.c_double, C.c_double]
        L.ref_mapper_get_variance_penalties.argtypes = [C.c_void_p, dptr, dptr]
        L.ref_matcher_create.restype = C.c_void_p
        L.ref_matcher_create.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.ref_matcher_destroy.argtypes = [C.c_void_p]
        L.ref_scan_create.restype = C.c_void_p
        L.ref_scan_create.argtypes = [dptr, C.c_int, dptr]
        L.ref_scan_destroy.argtypes = [C.c_void_p]
        L.ref_scan_set_pose.argtypes = [C.c_void_p, dptr]
        L.ref_scan_set_sensor_pose.argtypes = [C.c_void_p, dptr]
        L.ref_scan_get_sensor_pose.argtypes = [C.c_void_p, dptr]
        L.ref_scan_points.restype = C.c_int
        L.ref_scan_points.argtypes = [C.c_void_p, dptr, C.c_int]
        L.ref_find_valid_points.restype = C.c_int
        L.ref_find_valid_points.argtypes = [C.c_void_p, C.c_void_p, dptr, dptr, C.c_int]
        L.ref_match_scan.restype = C.c_double
        L.ref_match_scan.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, dptr, dptr]
        L.ref_add_scans.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int]
        L.ref_correlate_scan.restype = C.c_double
        L.ref_correlate_scan.argtypes = [C.c_void_p, C.c_void_p, dptr] + [C.c_double] * 6 + [C.c_int, C.c_int, dptr, dptr]
        L.ref_grid_info.argtypes = [C.c_void_p, iptr, dptr]
        L.ref_grid_data.restype = C.POINTER(C.c_uint8)
        L.ref_grid_data.argtypes = [C.c_void_p]
        L.ref_kernel_data.restype = C.POINTER(C.c_uint8)
        L.ref_kernel_data.argtypes = [C.c_void_p]
        L.ref_lookup_angles.restype = C.c_int
        L.ref_lookup_angles.argtypes = [C.c_void_p]
        L.ref_lookup_row.restype = C.c_int
        L.ref_lookup_row.argtypes = [C.c_void_p, C.c_int, iptr, C.c_int]
        L.ref_compute_offsets.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.ref_get_response.restype = C.c_double
        L.ref_get_response.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ref_world_to_grid_index.restype = C.c_int
        L.ref_world_to_grid_index.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_probs.restype = C.c_int
        L.ref_probs.argtypes = [C.c_void_p, dptr, C.c_int]
        L.ref_link_info.argtypes = [dptr, dptr, dptr, dptr, dptr]
        L.ref_matrix3_inverse.argtypes = [dptr, dptr]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class RefScan:
    def __init__(self, ranges, pose):
        self.n = len(ranges)
        self.h = lib().ref_scan_create(_d(ranges), self.n, _d(pose))

    def points(self):
        out = np.zeros((self.n, 2))
        n = lib().ref_scan_points(self.h, out, self.n)
        return out[:n]

    def sensor_pose(self):
        p = np.zeros(3)
        lib().ref_scan_get_sensor_pose(self.h, p)
        return p

    def set_sensor_pose(self, pose):
        lib().ref_scan_set_sensor_pose(self.h, _d(pose))

    def __del__(self):
        try:
            lib().ref_scan_destroy(self.h)
        except Exception:
            pass


class RefMatcher:
    """karto::ScanMatcher of the reference (Mapper.h:1322-1544) plus the Mapper that owns its parameters."""

    def __init__(self, search_size, resolution, smear, range_threshold, params=None):
        L = lib()
        self.mapper = L.ref_mapper_create()
        if params is not None:
            self.set_params(**params)
        self.h = L.ref_matcher_create(self.mapper, search_size, resolution, smear, range_threshold)
        if not self.h:
            raise ValueError("ScanMatcher::Create returned NULL / threw")

    def set_params(self, coarse_search_angle_offset, coarse_angle_resolution, fine_search_angle_offset,
                   use_response_expansion, distance_variance_penalty, minimum_distance_penalty,
                   angle_variance_penalty, minimum_angle_penalty):
        """distance/angle_variance_penalty are the values handed to the reference setters, which
        square them (Mapper.cpp:2562-2570)."""
        lib().ref_mapper_set_match_params(self.mapper, coarse_search_angle_offset, coarse_angle_resolution,
                                          fine_search_angle_offset, int(use_response_expansion),
                                          distance_variance_penalty, minimum_distance_penalty,
                                          angle_variance_penalty, minimum_angle_penalty)

    def _base(self, base):
        arr = (C.c_void_p * len(base))(*[b.h for b in base])
        return arr

    def match_scan(self, scan, base, do_penalize=True, do_refine=True):
        mean = np.zeros(3)
        cov = np.zeros(9)
        r = lib().ref_match_scan(self.h, scan.h, self._base(base), len(base), int(do_penalize), int(do_refine), mean, cov)
        return r, mean, cov.reshape(3, 3)

    def add_scans(self, scan, base):
        lib().ref_add_scans(self.h, scan.h, self._base(base), len(base))

    def correlate_scan(self, scan, center, off, res, ang_off, ang_res, do_penalize, fine, cov_in=None):
        mean = np.zeros(3)
        cov = np.zeros(9) if cov_in is None else _d(cov_in).reshape(9).copy()
        r = lib().ref_correlate_scan(self.h, scan.h, _d(center), off[0], off[1], res[0], res[1], ang_off, ang_res,
                                     int(do_penalize), int(fine), mean, cov)
        return r, mean, cov.reshape(3, 3)

    def grid_info(self):
        i = np.zeros(9, dtype=np.int32)
        d = np.zeros(3)
        lib().ref_grid_info(self.h, i, d)
        keys = ["width", "height", "width_step", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size", "data_size"]
        out = dict(zip(keys, (int(v) for v in i)))
        out.update(offset_x=d[0], offset_y=d[1], scale=d[2])
        return out

    def grid(self):
        info = self.grid_info()
        p = lib().ref_grid_data(self.h)
        return np.ctypeslib.as_array(p, shape=(info["data_size"],)).copy()

    def kernel(self):
        k = self.grid_info()["kernel_size"]
        p = lib().ref_kernel_data(self.h)
        return np.ctypeslib.as_array(p, shape=(k * k,)).copy().reshape(k, k)

    def compute_offsets(self, scan, angle_center, ang_off, ang_res):
        lib().ref_compute_offsets(self.h, scan.h, angle_center, ang_off, ang_res)

    def lookup_table(self, n_angles, n_points):
        out = np.zeros((n_angles, n_points), dtype=np.int32)
        for a in range(n_angles):
            row = np.zeros(n_points, dtype=np.int32)
            n = lib().ref_lookup_row(self.h, a, row, n_points)
            assert n == n_points
            out[a] = row
        return out

    def get_response(self, angle_index, grid_index):
        return lib().ref_get_response(self.h, angle_index, grid_index)

    def world_to_grid_index(self, x, y):
        return lib().ref_world_to_grid_index(self.h, x, y)

    def probs(self, side):
        out = np.zeros(side * side)
        n = lib().ref_probs(self.h, out, side * side)
        assert n == side * side
        return out.reshape(side, side)

    def find_valid_points(self, scan, viewpoint):
        out = np.zeros((scan.n, 2))
        n = lib().ref_find_valid_points(self.h, scan.h, _d(viewpoint), out, scan.n)
        return out[:n]

    def __del__(self):
        try:
            lib().ref_matcher_destroy(self.h)
            lib().ref_mapper_destroy(self.mapper)
        except Exception:
            pass


def init_laser(laser):
    n = lib().ref_init_laser(laser.min_angle, laser.max_angle, laser.ang_res, laser.min_range, laser.max_range,
                             laser.range_threshold)
    assert n == laser.n_beams, (n, laser.n_beams)
    return n


def link_info(pose1, pose2, cov):
    d = np.zeros(3)
    c = np.zeros(9)
    lib().ref_link_info(_d(pose1), _d(pose2), _d(cov).reshape(9), d, c)
    return d, c.reshape(3, 3)


def matrix3_inverse(a):
    out = np.zeros(9)
    lib().ref_matrix3_inverse(_d(a).reshape(9), out)
    return out.reshape(3, 3)
