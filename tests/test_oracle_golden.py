"""CPU: the plain-C oracle (oracle/karto_oracle.c) against the golden vectors generated from the
reference's own sources (tests/golden/make_golden.py).  Everything is compared bit-for-bit."""
import numpy as np
import pytest

from common import GOLDEN_NAMES, LASER, Golden, bits, make_oracle_matcher


@pytest.fixture(scope="module", params=GOLDEN_NAMES)
def golden(request):
    return Golden(request.param)


def test_points_and_valid_points(oracle_lib, golden):
    q, base = golden.oracle_scans()
    assert np.array_equal(bits(q.points), bits(golden.d["query_points"]))
    m = make_oracle_matcher(golden.preset)
    vp = m.find_valid_points(base[0], golden.query_pose[:2])
    assert np.array_equal(bits(vp), bits(golden.d["valid_points_0"]))


def test_geometry_and_kernel(oracle_lib, golden):
    m = make_oracle_matcher(golden.preset)
    gi = m.grid_info()
    geom = [gi[k] for k in ("width", "height", "width_step", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size", "data_size")]
    assert geom == [int(v) for v in golden.d["grid_geom"]]
    assert np.array_equal(m.kernel(), golden.d["kernel"])


def test_match_scan(oracle_lib, golden):
    q, base = golden.oracle_scans()
    m = make_oracle_matcher(golden.preset)
    for row, (pen, refine) in zip(golden.d["match_results"], [(True, True), (False, True), (False, False)]):
        r, mean, cov = m.match_scan(q, base, pen, refine)
        got = np.concatenate([[r], mean, cov.reshape(9)])
        assert np.array_equal(bits(got), bits(row)), (got, row)
    assert np.array_equal(m.grid(), golden.dense_grid())
    gi = m.grid_info()
    assert np.array_equal(bits([gi["offset_x"], gi["offset_y"], gi["scale"]]), bits(golden.d["grid_offset"]))


def test_correlate_scan(oracle_lib, golden):
    q, base = golden.oracle_scans()
    m = make_oracle_matcher(golden.preset, threads=4)
    m.add_scans(q, base)
    assert np.array_equal(m.grid(), golden.dense_grid())
    off, res, ang_off, ang_res, pen, fine = golden.correlate_args()
    r, mean, cov = m.correlate_scan(q, golden.query_pose, off, res, ang_off, ang_res, pen, fine)
    got = np.concatenate([[r], mean, cov.reshape(9)])
    assert np.array_equal(bits(got), bits(golden.d["correlate_result"]))
    assert np.array_equal(m.lookup_table(), golden.d["lookup"])
    if "probs" in golden.d.files:
        assert np.array_equal(bits(m.probs()), bits(golden.d["probs"]))
    raw = np.asarray([[m.get_response(a, int(c)) for a in range(golden.d["lookup"].shape[0])]
                      for c in golden.d["response_cells"]])
    assert np.array_equal(bits(raw), bits(golden.d["raw_responses"]))


def test_threads_do_not_change_results(oracle_lib):
    g = Golden("match_K")
    q, base = g.oracle_scans()
    a = make_oracle_matcher("K", threads=1).match_scan(q, base)
    b = make_oracle_matcher("K", threads=8).match_scan(q, base)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_create_rejects_bad_parameters(oracle_lib):
    from oracle import karto
    for args in [(0.5, 0.0, 0.1, 20.0), (0.0, 0.01, 0.1, 20.0), (0.5, 0.01, -1.0, 20.0), (0.5, 0.01, 0.1, 0.0),
                 (0.5, 0.01, 0.2, 20.0), (0.5, 0.01, 0.001, 20.0)]:   # last two: smear outside [0.5, 10] * res
        with pytest.raises(ValueError):
            karto.Matcher(*args)
