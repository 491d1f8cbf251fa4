"""Dev-container only: the C oracle against the reference's own karto_sdk build (oracle/_ref) on
fresh random scenarios, beyond the committed golden vectors.  Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from common import LASER, PRESETS, Scenario, bits, make_oracle_matcher
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module", autouse=True)
def _laser():
    ref.init_laser(LASER)


@pytest.mark.parametrize("preset,seed,start,n_base,perturb", [
    ("K", 31, 40, 5, (0.08, 0.02, -0.05)),
    ("K", 32, 150, 12, (-0.1, 0.1, 0.2)),
    ("S", 33, 77, 10, (0.02, 0.03, 0.01)),
    ("L", 34, 10, 30, (1.5, -2.0, 0.15)),
    ("L", 35, 300, 15, (-3.0, 0.5, -0.3)),
])
def test_match_scan_equals_reference(oracle_lib, preset, seed, start, n_base, perturb):
    sc = Scenario(seed=seed, n_base=n_base, start=start, perturb=perturb)
    rq, rb = sc.ref_scans()
    oq, ob = sc.oracle_scans()
    p = PRESETS[preset]
    mr = ref.RefMatcher(*p["create"], p["params"])
    mo = make_oracle_matcher(preset)
    for pen, refine in [(True, True), (False, False)]:
        a = mr.match_scan(rq, rb, pen, refine)
        b = mo.match_scan(oq, ob, pen, refine)
        assert np.array_equal(bits(a[0]), bits(b[0]))
        assert np.array_equal(bits(a[1]), bits(b[1]))
        assert np.array_equal(bits(a[2]), bits(b[2]))
        assert np.array_equal(mr.grid(), mo.grid())
        lt = mo.lookup_table()
        assert np.array_equal(mr.lookup_table(*lt.shape), lt)


def test_response_expansion_path(oracle_lib):
    """Query 90 degrees off: the coarse search finds nothing and the +20 degree expansion loop runs."""
    sc = Scenario(seed=36, n_base=6, start=50, perturb=(0.0, 0.0, 1.2))
    rq, rb = sc.ref_scans()
    oq, ob = sc.oracle_scans()
    p = PRESETS["S"]
    a = ref.RefMatcher(*p["create"], p["params"]).match_scan(rq, rb)
    b = make_oracle_matcher("S").match_scan(oq, ob)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))
    assert np.array_equal(bits(a[2]), bits(b[2]))


def test_empty_base_and_far_query(oracle_lib):
    sc = Scenario(seed=37, n_base=3, start=90, perturb=(25.0, 25.0, 0.0))
    rq, rb = sc.ref_scans()
    oq, ob = sc.oracle_scans()
    p = PRESETS["K"]
    mr = ref.RefMatcher(*p["create"], p["params"])
    mo = make_oracle_matcher("K")
    for base_r, base_o in [([], []), (rb, ob)]:
        a = mr.match_scan(rq, base_r)
        b = mo.match_scan(oq, base_o)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))
        assert np.array_equal(bits(a[2]), bits(b[2]))
