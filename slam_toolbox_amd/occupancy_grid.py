"""Python mirror of karto::OccupancyGrid (Karto.h:5880-6330) over the C ABI (kh_occupancy_*): same method
names and argument meaning as the reference -- CreateFromScans(scans, resolution), GetWidth/GetHeight,
cell states 0 unknown / 100 occupied / 255 free."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .scan_matcher import _scan_array

GridStates_Unknown, GridStates_Occupied, GridStates_Free = 0, 100, 255      # Karto.h:4379-4381


def compute_dimensions(scans, min_range, range_threshold, resolution):
    """OccupancyGrid::ComputeDimensions (Karto.h:6086-6112) -> (width, height, offset (2,))."""
    w, h = C.c_int32(), C.c_int32()
    off = np.zeros(2)
    capi.check(capi.lib().kh_occupancy_compute_dimensions(len(scans), _scan_array(scans), float(min_range),
                                                          float(range_threshold), float(resolution), C.byref(w),
                                                          C.byref(h), off), "kh_occupancy_compute_dimensions")
    return w.value, h.value, off


class OccupancyGrid:
    def __init__(self, width, height, offset, resolution, device: int = 0):
        h = C.c_void_p()
        capi.check(capi.lib().kh_occupancy_create(int(width), int(height), float(offset[0]), float(offset[1]),
                                                  float(resolution), device, C.byref(h)), "kh_occupancy_create")
        self._h = h
        self.width, self.height, self.width_step = int(width), int(height), (int(width) + 7) & ~7
        self.offset = np.asarray(offset, dtype=np.float64).copy()
        self.resolution = float(resolution)

    @staticmethod
    def CreateFromScans(scans, resolution, laser, device: int = 0, min_pass_through=2, occupancy_threshold=0.1):
        """OccupancyGrid::CreateFromScans (Karto.h:5947-5962); `laser` supplies GetRangeThreshold / GetMinimumRange /
        GetMaximumRange of the scans' LaserRangeFinder.  Returns None for an empty scan list like the reference."""
        if not scans:
            return None
        w, h, off = compute_dimensions(scans, laser.min_range, laser.range_threshold, resolution)
        g = OccupancyGrid(w, h, off, resolution, device)
        g.AddScans(scans, laser)
        g.Update(min_pass_through, occupancy_threshold)
        return g

    def AddScans(self, scans, laser):
        capi.check(capi.lib().kh_occupancy_add_scans(self._h, len(scans), _scan_array(scans), float(laser.range_threshold),
                                                     float(laser.min_range), float(laser.max_range)), "kh_occupancy_add_scans")

    def Update(self, min_pass_through=2, occupancy_threshold=0.1):
        capi.check(capi.lib().kh_occupancy_update(self._h, int(min_pass_through), float(occupancy_threshold)), "kh_occupancy_update")

    def Clear(self):
        capi.check(capi.lib().kh_occupancy_clear(self._h), "kh_occupancy_clear")

    def GetWidth(self):
        return self.width

    def GetHeight(self):
        return self.height

    def cells(self):
        out = np.zeros(self.width_step * self.height, dtype=np.uint8)
        capi.check(capi.lib().kh_occupancy_read(self._h, out.ctypes.data, None, None), "kh_occupancy_read")
        return out.reshape(self.height, self.width_step)

    def counters(self):
        p = np.zeros(self.width_step * self.height, dtype=np.uint32)
        h = np.zeros(self.width_step * self.height, dtype=np.uint32)
        capi.check(capi.lib().kh_occupancy_read(self._h, None, p.ctypes.data, h.ctypes.data), "kh_occupancy_read")
        return p.reshape(self.height, self.width_step), h.reshape(self.height, self.width_step)

    def stats(self):
        ms, beams = C.c_double(), C.c_int64()
        capi.check(capi.lib().kh_occupancy_info(self._h, None, None, None, C.byref(ms), C.byref(beams)), "kh_occupancy_info")
        return {"trace_ms": ms.value, "beams": beams.value}

    def close(self):
        if self._h:
            capi.lib().kh_occupancy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
