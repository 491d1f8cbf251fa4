"""GPU: the in-process multi-device candidate batch (kh_matcher_group_*, kh_mapper_create_on_devices).  One GPU is all the
tests have, so the device list names cuda:0 twice / three times: independent members with their own streams, workspaces
and host threads -- the N > 1 code path of SURVEY.md section 8e row A without RCCL.  The bar: results in candidate order,
bit-identical to the one-device calls (MapperGraph::TryCloseLoop's first-acceptance rule, Mapper.cpp:1500-1561, needs the
order; the matches themselves are independent)."""
import os

import numpy as np
import pytest

from common import PRESETS, Scenario, bits
from slam_toolbox_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group(preset, devices, max_batch):
    from slam_toolbox_amd.scan_matcher import MapperParams, ScanMatcherGroup
    p = PRESETS[preset]
    return ScanMatcherGroup(MapperParams(**p["params"]), *p["create"], devices=devices, max_batch_per_member=max_batch)


@pytest.mark.parametrize("preset,devices,max_batch", [("L", [0, 0], 8), ("S", [0, 0, 0], 3), ("K", [0], 16)])
def test_group_batch_equals_single_matcher(kartohip_lib, preset, devices, max_batch):
    from common import make_hip_matcher
    n = 13                                        # not a multiple of the member count, more than a member's chunk for S
    scs = [Scenario(seed=40 + i, n_base=5 + (i % 7), start=20 * i + 3, perturb=(0.02 * (i % 5), -0.015 * (i % 3), 0.004 * i)) for i in range(n)]
    qs, bs = [], []
    for sc in scs:
        q, b = sc.hip_scans()
        qs.append(q)
        bs.append(b)
    one = make_hip_matcher(preset, max_batch=n)
    r1, m1, c1, s1 = one.MatchScanBatch(qs, bs, False, preset != "L")
    g = _group(preset, devices, max_batch)
    assert len(g) == len(devices)
    r2, m2, c2, s2 = g.MatchScanBatch(qs, bs, False, preset != "L")
    assert (s1 == 0).all() and (s2 == 0).all()
    assert np.array_equal(bits(r1), bits(r2)) and np.array_equal(bits(m1), bits(m2)) and np.array_equal(bits(c1), bits(c2))
    # the same with the base scans resident on the members' device: a table of device copies, one column per member; every
    # member reads only its own column, scans missing from it are uploaded by the member
    flat = [b for lst in bs for b in lst]
    table = np.zeros((len(flat), len(devices)), dtype=np.uint64)
    row = 0
    for i, lst in enumerate(bs):
        for k, b in enumerate(lst):
            if (i + k) % 4 != 0:                  # leave a quarter of them out
                b.MakeResident()
                table[row, i % len(devices)] = b._resident.ptr
            row += 1
    r3, m3, c3, s3 = g.MatchScanBatch(qs, bs, False, preset != "L", device_points=table)
    assert np.array_equal(bits(r1), bits(r3)) and np.array_equal(bits(m1), bits(m3)) and np.array_equal(bits(c1), bits(c3))
    # empty batch, and a batch smaller than the group
    r4, _, _, _ = g.MatchScanBatch([], [])
    assert r4.shape == (0,)
    r5, m5, _, _ = g.MatchScanBatch(qs[:1], bs[:1], False, preset != "L")
    assert np.array_equal(bits(r5), bits(r1[:1])) and np.array_equal(bits(m5), bits(m1[:1]))
    one.close()
    g.close()


def test_mapper_on_two_members_runs_identically(kartohip_lib, tmp_path):
    """BASELINE config[0]'s queue (500 scans on the lap circuit, loops close) through the mapper front end with its
    candidate batches dealt over two members: the solver-call log -- every node, constraint, Compute and correction at 17
    digits -- and the final poses must be those of the one-device mapper."""
    from slam_toolbox_amd.mapper import Mapper
    n_scans = 500
    world = synth.make_world(12345)
    truth, odom = synth.trajectory_laps(n_scans)
    rng = np.random.default_rng(4)
    ranges = np.ascontiguousarray(np.stack([synth.make_scan(world, truth[i], rng) for i in range(n_scans)]))
    logs, poses, stats = [], [], []
    for devices in ([0], [0, 0]):
        log = str(tmp_path / f"m{len(devices)}.log")
        m = Mapper(synth.Laser(), loop_search_maximum_distance=3.0, log_path=log, devices=devices)
        for i in range(n_scans):
            m.Process(ranges[i], odom[i], 0.1 * i)
        poses.append(m.poses())
        stats.append(m.stats())
        m.set_log(None)
        m.close()
        # "X id ms": a Compute() with its wall time, "Z ...": timing notes -- the times are not part of the run
        logs.append([" ".join(l.split()[:2]) if l.startswith("X ") else l.rstrip("\n") for l in open(log) if not l.startswith("Z ")])
    assert stats[0]["loop_closures"] >= 1 and stats[0]["loop_closures"] == stats[1]["loop_closures"]
    # every sequential match of both runs took the fused path of one MatchScan (csrc/matcher_seq.cpp), its fine pass mostly
    # finished on the device (`matches` also counts the candidates of the loop-closure batches)
    for st in stats:
        assert st["fused_declined"] == 0 and st["fused_matches"] > 300 and st["fused_fine_passes"] > 0.8 * st["fused_matches"], st
    assert logs[0] == logs[1]
    assert np.array_equal(bits(poses[0]), bits(poses[1]))
