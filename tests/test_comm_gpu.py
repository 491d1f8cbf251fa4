"""GPU: the collectives inside the library (kh_comm_*: RCCL bound at run time).  One GPU is all the tests have:
  * a ONE-rank communicator runs ncclAllReduce / ncclAllGather for real (identity) -- the sharded solver with it must
    return bit-identical poses to the unsharded solver;
  * TWO ranks sharing cuda:0 are attempted too: RCCL may refuse two ranks on one device ("duplicate GPU"), in which case
    the test reports that and skips -- on a multi-GPU node the same worker runs with one device per rank."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_runs_the_collectives(kartohip_lib):
    # (no torch in this process: PyTorch ships its own HIP runtime, and whichever of the two runtimes initialises second
    # in one process finds no device; bench.py imports torch FIRST, after which libkartohip binds to torch's runtime)
    from slam_toolbox_amd import comm, synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    c = comm.Communicator(0, 0, 1, comm.unique_id())
    want = np.arange(1000) * 0.5
    x, y = comm.DeviceBuffer(1000), comm.DeviceBuffer(1000)
    x.upload(want)
    y.upload(np.zeros(1000))
    c.all_reduce_sum_f64(x.ptr, 1000)
    c.all_gather_f64(x.ptr, y.ptr, 1000)
    assert np.array_equal(x.download(), want) and np.array_equal(y.download(), want)
    x.free(); y.free()
    g = synth.make_pose_graph(1500, 4000, seed=13)
    a, b = HipSpaSolver(), HipSpaSolver()
    b.SetCommunicator(c)
    for s in (a, b):
        s.load(g["init"], g["edges"], g["z"], g["cov"])
    sa, sb = a.Compute(), b.Compute()
    assert sa["iterations"] == sb["iterations"] and sb["usable"] == 1
    assert np.array_equal(a.poses().view(np.uint64), b.poses().view(np.uint64))
    b.SetCommunicator(None)
    a.close(); b.close(); c.close()


def _worker(rank, world, id_path, out_dir):
    import sys
    import time
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from slam_toolbox_amd import capi, comm, synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    if rank == 0:
        uid = comm.unique_id()
        np.save(id_path + ".tmp.npy", uid)
        os.replace(id_path + ".tmp.npy", id_path)
    else:
        for _ in range(600):
            if os.path.exists(id_path):
                break
            time.sleep(0.05)
        uid = np.load(id_path)
    dev = rank % max(1, capi.lib().kh_device_count())
    try:
        c = comm.Communicator(dev, rank, world, uid)
    except capi.KartoHipError as exc:
        with open(os.path.join(out_dir, f"init_error_{rank}.txt"), "w") as f:
            f.write(str(exc))
        return
    g = synth.make_pose_graph(1500, 4000, seed=13)
    sol = HipSpaSolver(device=dev)
    sol.SetCommunicator(c)
    sol.load(g["init"], g["edges"], g["z"], g["cov"])
    summ = sol.Compute()
    np.save(os.path.join(out_dir, f"poses_{rank}.npy"), sol.poses())
    np.save(os.path.join(out_dir, f"iters_{rank}.npy"), np.asarray([summ["iterations"], summ["usable"]]))
    sol.close()
    c.close()


def test_two_rank_rccl_sharded_solve(kartohip_lib, tmp_path):
    import multiprocessing as mp
    from slam_toolbox_amd import synth
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    ctx = mp.get_context("spawn")
    id_path = str(tmp_path / "id.npy")
    procs = [ctx.Process(target=_worker, args=(r, 2, id_path, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    errs = [open(tmp_path / f"init_error_{r}.txt").read() for r in range(2) if os.path.exists(tmp_path / f"init_error_{r}.txt")]
    if errs or hung:
        # one device: RCCL refuses two ranks on it (ncclCommInitRank: invalid usage) -> nothing to test.  Two or more devices:
        # the ranks sit on different GPUs and a communicator that does not form is a FAILURE
        from slam_toolbox_amd import capi
        if capi.lib().kh_device_count() >= 2:
            pytest.fail(f"RCCL did not form a 2-rank communicator on {capi.lib().kh_device_count()} devices: {errs or 'init timed out'}")
        pytest.skip(f"RCCL would not form a 2-rank communicator on one device: {errs or 'init timed out'}")
    p0, p1 = np.load(tmp_path / "poses_0.npy"), np.load(tmp_path / "poses_1.npy")
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
    single.close()
