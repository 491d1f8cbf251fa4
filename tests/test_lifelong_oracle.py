"""CPU: the node-decay scoring oracle against the reference's own unit tests, test/lifelong_metrics_test.cpp:33-176
(the only tests slam_toolbox ships; same scans, same expected values and comparison modes)."""
import numpy as np

from oracle import lifelong


def _case():
    # bb1 = [2,5]x[2,6], barycenter (3.5, 4.0); bb2 = [2,5]x[4,7], barycenter (3.5, 5.5); two filtered points in s2
    s1 = lifelong.ScanBox(barycenter=(3.5, 4.0), bbox_size=(3.0, 4.0))
    s2 = lifelong.ScanBox(barycenter=(3.5, 5.5), bbox_size=(3.0, 3.0), points=np.array([[3.0, 5.0], [3.0, 3.0]]))
    return s1, s2


def test_bounds():                                 # TestBounds
    assert lifelong.intersect_bounds(*_case()) == (2.0, 5.0, 4.0, 6.0)


def test_intersect():                              # TestIntersect
    assert lifelong.intersect(*_case()) == 6.0


def test_intersect_over_union():                   # TestIntersectOverUnion (EXPECT_EQ)
    assert lifelong.intersect_over_union(*_case()) == 0.4


def test_area_overlap():                           # TestAreaOverlap (EXPECT_NEAR 0.6666, 0.01)
    assert abs(lifelong.area_overlap_ratio(*_case()) - 0.6666) <= 0.01


def test_reading_overlap():                        # TestPtOverlap
    assert lifelong.reading_overlap_ratio(*_case()) == 0.5
