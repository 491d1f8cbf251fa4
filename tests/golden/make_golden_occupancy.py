"""Generates tests/golden/occupancy.npz from the reference build (oracle/_ref/libkarto_ref.so): the reference's own
karto::OccupancyGrid::CreateFromScans (Karto.h:5947-5962) on 24 synthetic scans at 5 cm: grid geometry, the cell
states and both counter grids (stored sparsely).  Run in the dev container: python tests/golden/make_golden_occupancy.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ref  # noqa: E402
from slam_toolbox_amd import synth  # noqa: E402


def main():
    laser = synth.Laser()
    ref.init_laser(laser)
    L = ref.lib()
    world = synth.make_world(12345)
    truth, _ = synth.trajectory(400)
    rng = np.random.default_rng(21)
    idx = list(range(60, 60 + 24 * 3, 3))
    ranges = np.stack([synth.make_scan(world, truth[i], rng) for i in idx])
    poses = truth[idx] + rng.normal(0, 0.01, (len(idx), 3))
    scans = [ref.RefScan(ranges[k], poses[k]) for k in range(len(idx))]
    handles = (C.c_void_p * len(scans))(*[s.h for s in scans])
    dims = (C.c_int * 3)()
    off = (C.c_double * 2)()
    L.ref_occupancy_from_scans.restype = C.c_int
    L.ref_occupancy_from_scans.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    size = L.ref_occupancy_from_scans(handles, len(scans), 0.05, dims, off, None, None, None, 0)
    cells = np.zeros(size, dtype=np.uint8)
    passes = np.zeros(size, dtype=np.uint32)
    hits = np.zeros(size, dtype=np.uint32)
    L.ref_occupancy_from_scans(handles, len(scans), 0.05, dims, off, cells.ctypes.data, passes.ctypes.data, hits.ctypes.data, size)
    nz = np.flatnonzero(passes)
    out = os.path.join(ROOT, "tests", "golden", "occupancy.npz")
    np.savez_compressed(out, ranges=ranges, poses=poses, resolution=0.05, dims=np.asarray(list(dims), dtype=np.int32),
                        offset=np.asarray(list(off)), cells_idx=np.flatnonzero(cells).astype(np.int32),
                        cells_val=cells[np.flatnonzero(cells)], count_idx=nz.astype(np.int32), pass_val=passes[nz], hit_val=hits[nz])
    print(out, "grid", list(dims), "offset", list(off), "touched cells", nz.size, "occupied", int((cells == 100).sum()),
          "free", int((cells == 255).sum()))


if __name__ == "__main__":
    main()
