"""CPU, world_size 2 over gloo: the multi-GPU path of the matcher (slam_toolbox_amd/shard.py).  Each rank
matches its round-robin share of the candidate pairs -- here with the CPU oracle standing in for the GPU
matcher, which is what tests may do -- and every rank must end up with the same table as an unsharded run;
the timing reduction is the max over ranks."""
import os
import socket

import numpy as np
import pytest

from slam_toolbox_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _match_units(units):
    """Oracle MatchScan (karto default preset) of candidate pairs `units`: pair i = query scan at
    trajectory node 30 + 7 i against the 4 scans before it, pose perturbed deterministically."""
    from common import Scenario, make_oracle_matcher
    resp, means, covs = [], [], []
    m = make_oracle_matcher("K")
    for u in units:
        sc = Scenario(seed=100 + u, n_base=4, start=30 + 7 * u, perturb=(0.03 * ((u % 3) - 1), 0.02, 0.01 * (u % 2)))
        q, base = sc.oracle_scans()
        r, mean, cov = m.match_scan(q, base, False, True)
        resp.append(r); means.append(mean); covs.append(cov)
    return np.asarray(resp), np.asarray(means), np.asarray(covs)


def _worker(rank, world, port, n_units, out_dir):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = shard.match_candidates_sharded(_match_units, n_units, rank, world, batch=2)
    t = shard.max_over_ranks(1.0 + rank)
    np.save(os.path.join(out_dir, f"table_{rank}.npy"), table)
    np.save(os.path.join(out_dir, f"time_{rank}.npy"), np.asarray([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_units_partition():
    for n in (0, 1, 7, 256):
        for world in (1, 2, 8):
            parts = [shard.shard_units(n, r, world) for r in range(world)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard.shard_units(4, 2, 2)


def test_first_accepted_rule():
    t = np.zeros((4, 13))
    t[:, 0] = [0.2, 0.5, 0.6, 0.9]
    t[:, 4] = [1.0, 20.0, 2.0, 1.0]        # cov(0,0)
    t[:, 8] = [1.0, 1.0, 2.0, 1.0]         # cov(1,1)
    assert shard.first_accepted(t, 0.35, 9.0) == 2        # 0: response too low, 1: variance too high
    assert shard.first_accepted(t[:2], 0.35, 9.0) == -1


def test_two_ranks_match_the_unsharded_run(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    n_units = 5
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_units, str(tmp_path)), nprocs=2, join=True)
    t0 = np.load(tmp_path / "table_0.npy")
    t1 = np.load(tmp_path / "table_1.npy")
    assert np.array_equal(t0.view(np.uint64), t1.view(np.uint64))          # every rank has the same table
    resp, means, covs = _match_units(list(range(n_units)))
    ref = np.concatenate([resp.reshape(-1, 1), means.reshape(-1, 3), covs.reshape(-1, 9)], axis=1)
    assert np.array_equal(t0.view(np.uint64), ref.view(np.uint64))        # ... equal to the 1-rank result, bit for bit
    assert np.load(tmp_path / "time_0.npy")[0] == 2.0 and np.load(tmp_path / "time_1.npy")[0] == 2.0
