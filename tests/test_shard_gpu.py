"""GPU: the sharded candidate matcher (slam_toolbox_amd/shard.py) with the HIP matcher doing
the matching and DEVICE tensors in the gather.  One GPU is all the tests have, so two ranks share cuda:0 and the
fixed-size all-gather runs over gloo (which takes device tensors); with backend "nccl" the same lines are RCCL over xGMI.
Every rank must end up with the table an unsharded HIP run produces, bit for bit, and with the same first accepted
candidate (TryCloseLoop's rule, Mapper.cpp:1515-1518)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_UNITS = 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _matcher_and_pairs():
    from common import LASER, OFFLINE_PARAMS, PRESETS
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_matcher import LocalizedRangeScan, MapperParams, ScanMatcher
    lb = synth.loop_batch(N_UNITS)
    cache = {}

    def at(i):
        if i not in cache:
            cache[i] = LocalizedRangeScan(lb["ranges"][i], lb["truth"][i], LASER.min_angle, LASER.ang_res)
        return cache[i]
    queries = [LocalizedRangeScan(lb["ranges"][q], pose, LASER.min_angle, LASER.ang_res) for q, pose, _ in lb["pairs"]]
    chains = [[at(i) for i in chain] for _, _, chain in lb["pairs"]]
    m = ScanMatcher.Create(MapperParams(**OFFLINE_PARAMS), *PRESETS["L"]["create"], max_batch=8)

    def match_fn(units):
        resp, means, covs, st = m.MatchScanBatch([queries[u] for u in units], [chains[u] for u in units], False, False)
        assert (st == 0).all()
        return resp, means, covs
    return m, match_fn


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import torch                                  # before libkartohip: the library then binds to torch's HIP runtime
    import torch.distributed as dist
    from slam_toolbox_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, match_fn = _matcher_and_pairs()
    table = shard.match_candidates_sharded(match_fn, N_UNITS, rank, world, batch=8, device="cuda")
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table)
    m.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_the_hip_matcher_equal_the_unsharded_run(kartohip_lib, tmp_path):
    import torch.multiprocessing as mp
    from slam_toolbox_amd import shard
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "table_0.npy"), np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0.view(np.uint64), t1.view(np.uint64))
    m, match_fn = _matcher_and_pairs()
    rows = []
    for b in range(0, N_UNITS, 8):
        resp, means, covs = match_fn(list(range(b, min(N_UNITS, b + 8))))
        rows.append(np.concatenate([resp.reshape(-1, 1), means.reshape(-1, 3), covs.reshape(-1, 9)], axis=1))
    want = np.concatenate(rows, axis=0)
    m.close()
    assert np.array_equal(t0.view(np.uint64), want.view(np.uint64))
    assert shard.first_accepted(t0, 0.35, 9.0) == shard.first_accepted(want, 0.35, 9.0) >= 0
