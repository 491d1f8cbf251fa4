"""Python mirror of the node-decay scoring of slam_toolbox::LifelongSlamToolbox (computeScores and the metrics,
src/experimental/slam_toolbox_lifelong.cpp:199-329, 373-478) over the C ABI (kh_lifelong_scores)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _box(s, keep):
    b = capi.KhScanBox()
    b.barycenter[0], b.barycenter[1] = float(s.barycenter[0]), float(s.barycenter[1])
    b.bbox_size[0], b.bbox_size[1] = float(s.bbox_size[0]), float(s.bbox_size[1])
    b.unique_id, b.n_edges, b.score = int(s.unique_id), int(s.n_edges), float(s.score)
    pts = np.ascontiguousarray(np.asarray(s.points, dtype=np.float64).reshape(-1, 2))
    keep.append(pts)
    b.n_points = pts.shape[0]
    b.points_xy = pts.ctypes.data_as(C.POINTER(C.c_double))
    return b


def computeScores(reference, candidates, params=None, device: int = 0):
    """-> (kept, iou, area_overlap, reading_overlap, score) arrays over `candidates`.  `reference` / candidates are
    objects with barycenter, bbox_size, points (filtered readings), unique_id, n_edges, score; `params` any object
    with the kh_decay_params field names (defaults: slam_toolbox_lifelong.cpp:60-100)."""
    p = capi.KhDecayParams()
    capi.lib().kh_decay_params_default(C.byref(p))
    if params is not None:
        for name, _ in capi.KhDecayParams._fields_:
            if hasattr(params, name):
                setattr(p, name, getattr(params, name))
    keep = []
    ref = _box(reference, keep)
    n = len(candidates)
    arr = (capi.KhScanBox * max(n, 1))(*[_box(c, keep) for c in candidates])
    kept = np.zeros(n, dtype=np.int32)
    iou, area, reading, score = (np.zeros(n) for _ in range(4))
    capi.check(capi.lib().kh_lifelong_scores(device, C.byref(ref), n, arr, C.byref(p), kept.ctypes.data, iou.ctypes.data,
                                             area.ctypes.data, reading.ctypes.data, score.ctypes.data), "kh_lifelong_scores")
    return kept.astype(bool), iou, area, reading, score
