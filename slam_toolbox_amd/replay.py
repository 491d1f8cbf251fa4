"""BASELINE config 5: lifelong-mode replay of a synthetic warehouse bag, end to end -- scan queue -> mapper front end
(kh_mapper_*: sequential match, links, speculative loop closure, SPA solves, node decay) -> occupancy grid of the final
graph -- with the two queue policies of slam_toolbox:

  sync   every scan is processed (sync_slam_toolbox_node / the offline launch)
  async  AsynchronousSlamToolbox::laserCallback (src/slam_toolbox_async.cpp:34-57) behind a depth-1 subscription: scans
         arrive every `period_s` of wall time; while the mapper is busy the newer scan replaces the waiting one, i.e. on
         becoming free the mapper takes the most recent scan that has arrived and drops the ones before it.

Nothing here computes a match or a pose: the module builds the queue and calls the library."""
from __future__ import annotations

import math
import time

import numpy as np

from . import synth


class LapQueue:
    """n_scans scans of the lap circuit (synth.trajectory_laps).  The circuit is periodic, so the exact ray casting is done
    once per circuit position and every scan adds its own range noise, +inf and NaN beams on top."""

    # Odometry noise per 0.5 m step: 5 mm in x and y (1 %) and 0.1 deg (0.2 deg/m) -- ordinary wheel odometry.  The round-2
    # queue used 2 cm and 0.5 deg per step (4 %, 1 deg/m): with that the REFERENCE mapper itself ends 0.24 m rms / 1.2 m max
    # from the ground truth after 1500 scans (tools/replay_vs_reference.py: reference and library poses identical bit for
    # bit), so the map's agreement with the truth said nothing about this library.  With this noise both end at 2.3 cm rms.
    def __init__(self, n_scans: int, seed: int = 4, aisles=(0, 1), drift_xy: float = 0.005, drift_theta_deg: float = 0.1):
        self.laser = synth.Laser()
        self.world = synth.make_world(12345)
        self.truth, self.odom = synth.trajectory_laps(n_scans, aisles=aisles, drift_xy=drift_xy, drift_theta_deg=drift_theta_deg)
        self.rng = np.random.default_rng(seed)
        self._cast = {}
        self.n = n_scans

    def ranges(self, i: int) -> np.ndarray:
        key = tuple(np.round(self.truth[i], 9))
        if key not in self._cast:
            self._cast[key] = synth.raycast(self.world, self.truth[i], self.laser)
        r = self._cast[key] + self.rng.normal(0.0, 0.01, size=self.laser.n_beams)
        r = np.where(r > self.laser.max_range, self.laser.max_range, r)
        r = np.maximum(r, 0.02)
        flags = self.rng.uniform(size=r.shape)
        r = np.where(flags < 0.01, np.inf, r)
        r = np.where((flags >= 0.01) & (flags < 0.015), np.nan, r)
        return r


def run(n_scans: int, lifelong: bool = True, mode: str = "sync", period_s: float = 0.025, device: int = 0,
        max_candidates: int = 32, map_resolution: float = 0.05, progress=None, devices=None, queue=None, **mapper_params):
    """Replays the queue; returns a dict with the throughput, the mapper's own statistics and the map agreement."""
    from .mapper import Mapper
    from .occupancy_grid import OccupancyGrid
    from .scan_matcher import LocalizedRangeScan
    q = queue or LapQueue(n_scans)
    m = Mapper(q.laser, device=device, max_candidates=max_candidates, devices=devices, **mapper_params)
    if lifelong:
        m.SetLifelong(True)
    # the queue is made up front so that the timed region holds the mapper, not the ray casting
    all_ranges = [q.ranges(i) for i in range(n_scans)]
    t0 = time.perf_counter()
    processed = dropped = accepted = 0
    waited_s = 0.0
    queue_index = []                       # scan id -> position in the queue
    i = 0
    while i < n_scans:
        if mode == "async":
            # scan k ARRIVES at k * period_s.  The depth-1 queue holds the most recent arrival: a mapper that is behind
            # skips to it (the ones in between are dropped), a mapper that is ahead waits for the next scan to arrive --
            # with a mapper faster than the sensor nothing is dropped and the run lasts n_scans * period_s
            now = time.perf_counter() - t0
            latest = min(n_scans - 1, int(now / period_s))
            if latest > i:
                dropped += latest - i
                i = latest
            elif now < i * period_s:
                time.sleep(i * period_s - now)
                waited_s += i * period_s - now
        ok, _, _ = m.Process(all_ranges[i], q.odom[i], 0.1 * i)
        processed += 1
        if ok:
            accepted += 1
            queue_index.append(i)
        i += 1
        if progress and processed % progress == 0:
            st = m.stats()
            print(f"[replay] {i}/{n_scans} scans, {accepted} accepted, {len(m.alive())} alive, {st['loop_closures']} closures, "
                  f"{time.perf_counter() - t0:.1f} s", flush=True)
    wall = time.perf_counter() - t0
    st = m.stats()
    alive = m.alive()
    # connected components of the pose graph that is left (node decay removes vertices with their edges: nothing in
    # LifelongSlamToolbox::evaluateNodeDepreciation keeps the graph connected)
    comp_of = {}
    for root in alive.tolist():
        if root in comp_of:
            continue
        comp_of[root] = root
        todo = [root]
        while todo:
            v = todo.pop()
            for u in m.adjacency(v).tolist():
                if u not in comp_of:
                    comp_of[u] = root
                    todo.append(u)
    comp_sizes = sorted(np.unique(list(comp_of.values()), return_counts=True)[1].tolist(), reverse=True) if comp_of else []
    poses = m.poses()[alive]
    truth = q.truth[[queue_index[k] for k in alive]]
    # final map: OccupancyGrid::CreateFromScans over the scans still in the graph at their corrected poses, against the map
    # the same scans draw from their TRUE poses on the same cells (the mapper's frame is the odometry frame of the first
    # scan = the truth there): intersection over union of the occupied cells, and the pose error of the alive nodes
    t1 = time.perf_counter()
    rng_of = [all_ranges[queue_index[k]] for k in alive]
    true_scans = [LocalizedRangeScan(r, p, q.laser.min_angle, q.laser.ang_res) for r, p in zip(rng_of, truth)]
    scans = [LocalizedRangeScan(r, p, q.laser.min_angle, q.laser.ang_res) for r, p in zip(rng_of, poses)]
    t2 = time.perf_counter()
    ref_grid = OccupancyGrid.CreateFromScans(true_scans + scans, map_resolution, q.laser, device)      # dimensions that hold both
    ref_grid.Clear()
    ref_grid.AddScans(true_scans, q.laser)
    ref_grid.Update()
    want = ref_grid.cells()[:, :ref_grid.width].copy()
    t3 = time.perf_counter()
    ref_grid.Clear()
    ref_grid.AddScans(scans, q.laser)
    ref_grid.Update()
    map_ms = (time.perf_counter() - t3) * 1e3
    cells = ref_grid.cells()[:, :ref_grid.width]
    occ_a, occ_b = cells == 100, want == 100
    iou = float((occ_a & occ_b).sum()) / max(1, int((occ_a | occ_b).sum()))
    # occupied cells within one cell of an occupied cell of the other map (5 cm of slack on 1 cm range noise)
    def dilate(x):
        y = x.copy()
        y[1:] |= x[:-1]; y[:-1] |= x[1:]; y[:, 1:] |= x[:, :-1]; y[:, :-1] |= x[:, 1:]
        return y
    near = float((occ_a & dilate(occ_b)).sum()) / max(1, int(occ_a.sum()))
    d = poses - truth
    d[:, 2] = (d[:, 2] + math.pi) % (2 * math.pi) - math.pi
    # the same after the best rigid motion of the whole map onto the truth (2-D Procrustes over the node positions): what
    # is left is the map's internal distortion, without the heading the first metres of odometry gave the frame
    pc, tc = poses[:, :2].mean(0), truth[:, :2].mean(0)
    H = (poses[:, :2] - pc).T @ (truth[:, :2] - tc)
    ang = math.atan2(H[0, 1] - H[1, 0], H[0, 0] + H[1, 1])
    R = np.array([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])
    aligned = poses.copy()
    aligned[:, :2] = (poses[:, :2] - pc) @ R.T + tc
    aligned[:, 2] = poses[:, 2] + ang
    da = aligned - truth
    da[:, 2] = (da[:, 2] + math.pi) % (2 * math.pi) - math.pi
    ref_grid.Clear()
    ref_grid.AddScans([LocalizedRangeScan(r, p, q.laser.min_angle, q.laser.ang_res) for r, p in zip(rng_of, aligned)], q.laser)
    ref_grid.Update()
    occ_c = ref_grid.cells()[:, :ref_grid.width] == 100
    iou_aligned = float((occ_c & occ_b).sum()) / max(1, int((occ_c | occ_b).sum()))
    near_aligned = float((occ_c & dilate(occ_b)).sum()) / max(1, int(occ_c.sum()))
    out = {"scans": n_scans, "processed": processed, "dropped": dropped, "accepted": accepted, "alive": int(len(alive)),
           "wall_s": wall, "scans_per_s": processed / wall, "mode": mode, "lifelong": lifelong,
           "graph_components": len(comp_sizes), "graph_largest_components": comp_sizes[:5],
           "period_s": period_s if mode == "async" else None, "waited_for_arrivals_s": waited_s, "busy_s": wall - waited_s,
           "map_build_ms": map_ms, "map_cells": [int(ref_grid.width), int(ref_grid.height)],
           "map_occupied": int(occ_a.sum()), "map_free": int((cells == 255).sum()),
           "map_iou_vs_truth_poses": iou, "map_occupied_within_one_cell_of_truth_map": near,
           "pose_error_xy_max_m": float(np.abs(d[:, :2]).max()), "pose_error_xy_rms_m": float(np.sqrt((d[:, :2] ** 2).sum(1).mean())),
           "pose_error_heading_max_rad": float(np.abs(d[:, 2]).max()),
           "aligned": {"rotation_rad": ang, "map_iou_vs_truth_poses": iou_aligned,
                       "map_occupied_within_one_cell_of_truth_map": near_aligned,
                       "pose_error_xy_max_m": float(np.abs(da[:, :2]).max()),
                       "pose_error_xy_rms_m": float(np.sqrt((da[:, :2] ** 2).sum(1).mean())),
                       "pose_error_heading_max_rad": float(np.abs(da[:, 2]).max())},
           "stats": st}
    # where the wall time of the replay loop went: inside kh_mapper_process (the mapper's own split; `other_in_process` = scan update,
    # graph store sync, candidate enumeration, links) and outside it (this loop, ctypes marshalling of 1081 ranges per scan)
    known = st["match_ms"] + st["solver_ms"] + st["update_ms"] + st["lifelong_ms"]
    out["ms_split"] = {"match": st["match_ms"], "solver": st["solver_ms"], "pose_updates": st["update_ms"], "node_decay": st["lifelong_ms"],
                       "other_in_process": st["process_ms"] - known, "process_total": st["process_ms"],
                       "outside_process_python_ctypes": (wall - waited_s) * 1e3 - st["process_ms"], "wall": (wall - waited_s) * 1e3}
    out["poses"] = poses
    out["alive_queue_index"] = [queue_index[k] for k in alive]
    ref_grid.close()
    m.close()
    return out
