"""GPU: the multi-GPU path of the solver -- edge-block sharded linearisation,
partial normal equations summed by torch.distributed.all_reduce through the C-ABI callback.  Only one GPU is
available to the tests, so two ranks share cuda:0 and the collective runs over gloo (which accepts device
tensors); with backend "nccl" the same code path is RCCL over xGMI."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    import torch
    import torch.distributed as dist
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = synth.make_pose_graph(1500, 4000, seed=13)
    sol = HipSpaSolver()
    sol.enable_sharding(rank, world)
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    np.save(os.path.join(out_dir, f"poses_{rank}.npy"), sol.poses())
    np.save(os.path.join(out_dir, f"iters_{rank}.npy"), np.asarray([summ["iterations"], summ["usable"]]))
    with open(os.path.join(out_dir, f"warn_{rank}.txt"), "w") as f:
        f.write(sol.last_warning)
    sol.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_linearisation_matches_single_gpu(kartohip_lib, tmp_path):
    import torch.multiprocessing as mp
    from oracle import spa
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "poses_0.npy"), np.load(tmp_path / "poses_1.npy")
    assert open(tmp_path / "warn_0.txt").read() == "" and open(tmp_path / "warn_1.txt").read() == ""
    assert np.array_equal(p0.view(np.uint64), p1.view(np.uint64))     # replicated solve: identical on every rank
    g = synth.make_pose_graph(1500, 4000, seed=13)
    single = HipSpaSolver()
    single.load(g["init"], g["edges"], g["z"], g["cov"])
    s1 = single.Compute()
    it = np.load(tmp_path / "iters_0.npy")
    assert it[1] == 1 and it[0] == s1["iterations"]
    d = p0 - single.poses()
    d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-7                                     # the partial sums only re-associate H and g
    ref_x, info = spa.solve(g["init"], g["edges"], g["z"], g["cov"])
    d = p0 - ref_x
    d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-7
    single.close()
