"""Host-side mirror of karto::Mapper's processing entry point (lib/karto_sdk/src/Mapper.cpp:2679-2748) over the mapper
front end of libkartohip.so (kh_mapper_*): a ROS-free way to replay a scan queue end to end on the GPU.

    mapper = Mapper(laser, loop_search_maximum_distance=3.0)        # parameters of config/mapper_params_offline.yaml
    accepted, pose, cov = mapper.Process(ranges, odometric_pose, time)
    poses = mapper.poses()

Nothing here computes: every call lands in the library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Mapper:
    def __init__(self, laser, device: int = 0, max_candidates: int = 32, log_path: str = None, devices=None, **params):
        """laser: anything with n_beams, min_angle, ang_res, min_range, max_range, range_threshold (synth.Laser);
        params: fields of kh_mapper_params to override (AS STORED by karto::Mapper: variances squared);
        devices: device list of kh_mapper_create_on_devices (candidate batches dealt over one matcher pair per entry; the
        same device may be listed more than once), default [device]."""
        p = capi.KhMapperParams()
        capi.lib().kh_mapper_params_default(C.byref(p))
        match_fields = {k for k, _ in capi.KhMatchParams._fields_}
        for k, v in params.items():
            if k in match_fields:
                setattr(p.match, k, v)
            elif hasattr(p, k):
                setattr(p, k, v)
            else:
                raise KeyError(k)
        off = tuple(getattr(laser, "offset", (0.0, 0.0, 0.0)))      # where the sensor sits on the robot (x, y, heading)
        L = capi.KhLaser(laser.n_beams, laser.min_angle, laser.ang_res, laser.min_range, laser.max_range, laser.range_threshold,
                         off[0], off[1], off[2])
        self._h = C.c_void_p()
        devs = np.asarray([device] if devices is None else list(devices), dtype=np.int32)
        capi.check(capi.lib().kh_mapper_create_on_devices(C.byref(p), C.byref(L), devs, len(devs), max_candidates, C.byref(self._h)),
                   "kh_mapper_create_on_devices")
        self.n_beams = laser.n_beams
        if log_path:
            capi.check(capi.lib().kh_mapper_set_log(self._h, log_path.encode()), "kh_mapper_set_log")

    def Process(self, ranges, odometric_pose, time: float = 0.0):
        ranges = np.ascontiguousarray(ranges, dtype=np.float64)
        assert ranges.shape == (self.n_beams,)
        acc = C.c_int32(0)
        pose, cov = np.zeros(3), np.zeros(9)
        capi.check(capi.lib().kh_mapper_process(self._h, ranges, np.ascontiguousarray(odometric_pose, dtype=np.float64), float(time),
                                                C.byref(acc), pose, cov), "kh_mapper_process")
        return bool(acc.value), pose, cov.reshape(3, 3)

    def num_scans(self) -> int:
        return capi.lib().kh_mapper_num_scans(self._h)

    def num_edges(self) -> int:
        return capi.lib().kh_mapper_num_edges(self._h)

    def poses(self) -> np.ndarray:
        out = np.zeros((self.num_scans(), 3))
        if out.size:
            capi.check(capi.lib().kh_mapper_get_poses(self._h, out.reshape(-1)), "kh_mapper_get_poses")
        return out

    def scan(self, index: int):
        """(kh_scan, kh_scan_box) views of scan `index`: what the occupancy grid and the lifelong scoring read"""
        s, b = capi.KhScan(), capi.KhScanBox()
        capi.check(capi.lib().kh_mapper_get_scan(self._h, index, C.byref(s), C.byref(b)), "kh_mapper_get_scan")
        return s, b

    def adjacency(self, scan_id: int) -> np.ndarray:
        """Vertex::GetAdjacentVertices of the scan's vertex, in the reference's order"""
        n = C.c_int32(0)
        out = np.zeros(64, dtype=np.int32)
        capi.check(capi.lib().kh_mapper_get_adjacency(self._h, int(scan_id), out, out.size, C.byref(n)), "kh_mapper_get_adjacency")
        if n.value > out.size:
            out = np.zeros(n.value, dtype=np.int32)
            capi.check(capi.lib().kh_mapper_get_adjacency(self._h, int(scan_id), out, out.size, C.byref(n)), "kh_mapper_get_adjacency")
        return out[:n.value]

    def SetNodeScore(self, scan_id: int, score: float):
        """Vertex::SetScore (LifelongSlamToolbox::updateScoresSlamGraph)"""
        capi.check(capi.lib().kh_mapper_set_node_score(self._h, int(scan_id), float(score)), "kh_mapper_set_node_score")

    def RemoveNode(self, scan_id: int):
        """Mapper::RemoveNodeFromGraph + MapperSensorManager::RemoveScan (what lifelong mode does to a decayed node)"""
        capi.check(capi.lib().kh_mapper_remove_node(self._h, int(scan_id)), "kh_mapper_remove_node")

    def SetLifelong(self, enabled: bool = True, **decay):
        """LifelongSlamToolbox::evaluateNodeDepreciation after every accepted scan; decay = kh_decay_params overrides"""
        if not enabled:
            capi.check(capi.lib().kh_mapper_set_lifelong(self._h, None), "kh_mapper_set_lifelong")
            return
        p = capi.KhDecayParams()
        capi.lib().kh_decay_params_default(C.byref(p))
        for k, v in decay.items():
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
        capi.check(capi.lib().kh_mapper_set_lifelong(self._h, C.byref(p)), "kh_mapper_set_lifelong")

    def alive(self) -> np.ndarray:
        ids = np.zeros(max(1, capi.lib().kh_mapper_num_alive(self._h)), dtype=np.int32)
        capi.check(capi.lib().kh_mapper_get_alive(self._h, ids), "kh_mapper_get_alive")
        return ids[:capi.lib().kh_mapper_num_alive(self._h)]

    def stats(self) -> dict:
        st = capi.KhMapperStats()
        capi.check(capi.lib().kh_mapper_get_stats(self._h, C.byref(st)), "kh_mapper_get_stats")
        return {k: getattr(st, k) for k, _ in capi.KhMapperStats._fields_}

    def set_log(self, path):
        capi.check(capi.lib().kh_mapper_set_log(self._h, path.encode() if path else None), "kh_mapper_set_log")

    def close(self):
        if self._h:
            capi.lib().kh_mapper_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
