"""GPU box (the solver handle needs a device): kh_spa_save / kh_spa_load / kh_spa_add_constraint_information
through the C ABI against the CPU restatement oracle/posegraph.py and the committed golden files."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _state(sol):
    nodes, cons = sol.nodes_in_order(), sol.constraints_in_order()
    return dict(ids=np.asarray([i for i, _ in nodes], dtype=np.int32), poses=np.asarray([p for _, p in nodes]).reshape(-1, 3),
                edges=np.asarray([[a, b] for a, b, _, _ in cons], dtype=np.int32).reshape(-1, 2),
                z=np.asarray([z for _, _, z, _ in cons]).reshape(-1, 3), info=np.asarray([w for _, _, _, w in cons]).reshape(-1, 6))


def _same(got, want):
    for k in ("ids", "poses", "edges", "z", "info"):
        assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), k


@pytest.mark.parametrize("name", ["posegraph_small.g2o", "posegraph_small.khpg"])
def test_load_golden_files(kartohip_lib, name):
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    want = np.load(os.path.join(GOLD, "posegraph_small.npz"))
    sol = HipSpaSolver()
    sol.AddNode(999, [1.0, 2.0, 3.0])                   # load_graph resets first (slam_toolbox_common.cpp:959)
    sol.load_graph(os.path.join(GOLD, name))
    _same(_state(sol), want)
    sol.close()


def test_save_round_trip_and_oracle_reads_it(kartohip_lib, tmp_path):
    from oracle import posegraph
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    want = np.load(os.path.join(GOLD, "posegraph_small.npz"))
    sol = HipSpaSolver()
    # build through the plugin API: AddNode / AddConstraint with LinkInfo covariances (ceres_solver.cpp:339-392)
    for i, p in zip(want["ids"], want["poses"]):
        sol.AddNode(int(i), p)
    for (a, b), z, c in zip(want["edges"], want["z"], want["cov"]):
        sol.AddConstraint(int(a), int(b), z, c.reshape(3, 3))
    _same(_state(sol), want)                           # information = the oracle's symmetrised inverse, bit for bit
    t, b = str(tmp_path / "s.g2o"), str(tmp_path / "s.khpg")
    sol.save_graph(t)
    sol.save_graph(b, binary=True)
    got = posegraph.read_text(t)
    _same(got, want)
    assert got["fix"] == int(want["ids"][0])
    with open(b, "rb") as f, open(os.path.join(GOLD, "posegraph_small.khpg"), "rb") as g:
        assert f.read() == g.read()
    other = HipSpaSolver()
    other.load_graph(t)
    _same(_state(other), want)
    # same graph, same solve: a loaded graph optimises to exactly the poses of the one built through AddConstraint
    sol._ids = other._ids = [int(i) for i in want["ids"]]
    s1, s2 = sol.Compute(), other.Compute()
    assert s1["usable"] == 1 and s1["iterations"] == s2["iterations"]
    assert np.array_equal(sol.poses(), other.poses())
    # and the optimised graph round-trips too
    sol.save_graph(t)
    other.load_graph(t)
    assert np.array_equal(sol.poses(), other.poses())
    sol.close(); other.close()


def test_g2o_layout_sample_loads_and_solves(kartohip_lib):
    from oracle import posegraph, spa
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    path = os.path.join(GOLD, "posegraph_g2o_sample.g2o")
    want = posegraph.read_text(path)
    sol = HipSpaSolver()
    sol.load_graph(path)
    _same(_state(sol), want)
    summ = sol.Compute()
    assert summ["usable"] == 1
    # the oracle solves from covariances: hand it the inverse of the file's information
    cov = []
    for w in want["info"]:
        cov.append(np.linalg.inv(np.array([[w[0], w[1], w[2]], [w[1], w[3], w[4]], [w[2], w[4], w[5]]])).reshape(9))
    ref_x, info = spa.solve(want["poses"], want["edges"], want["z"], np.asarray(cov))
    assert summ["iterations"] == info["iterations"]
    d = sol.poses() - ref_x
    assert np.abs(d).max() < 1e-7
    sol.close()


def test_bad_files_leave_the_graph_alone(kartohip_lib, tmp_path):
    from slam_toolbox_amd import capi
    from slam_toolbox_amd.scan_solver import HipSpaSolver
    sol = HipSpaSolver()
    sol.load_graph(os.path.join(GOLD, "posegraph_g2o_sample.g2o"))
    before = _state(sol)

    def expect(code, text=None, data=None):
        p = str(tmp_path / "bad")
        if text is not None:
            with open(p, "w") as f:
                f.write(text)
        if data is not None:
            with open(p, "wb") as f:
                f.write(data)
        with pytest.raises(capi.KartoHipError) as e:
            sol.load_graph(p)
        assert e.value.code == code, str(e.value)
        _same(_state(sol), before)

    with pytest.raises(capi.KartoHipError) as e:
        sol.load_graph(str(tmp_path / "does_not_exist.g2o"))
    assert e.value.code == capi.KH_ERR_IO
    _same(_state(sol), before)
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0\n")                                   # short record
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 0 7\n")                               # trailing field
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 zero\n")                              # not a number
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_XY 0 0 0\n")                                    # unsupported record
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 0\nVERTEX_SE2 0 1 0 0\n")             # duplicate id
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 0\nEDGE_SE2 0 1 1 0 0 1 0 0 1 0 1\n")  # unknown vertex
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0 0\nFIX 1\n")      # gauge is the first node
    expect(capi.KH_ERR_INVALID_ARG, text="VERTEX_SE2 0 0 0 0\nVERTEX_SE2 1 1 0 0\nEDGE_SE2 0 1 1 0 0 1 2 0 1 0 1\n")  # not PD
    with open(os.path.join(GOLD, "posegraph_small.khpg"), "rb") as f:
        blob = f.read()
    expect(capi.KH_ERR_INVALID_ARG, data=blob[:-5])                                              # truncated binary
    # an empty file is an empty graph
    p = str(tmp_path / "empty.g2o")
    open(p, "w").close()
    sol.load_graph(p)
    assert capi.lib().kh_spa_num_nodes(sol._h) == 0
    sol.close()
