"""GPU parity of the lifelong node-decay scoring (kh_lifelong_scores, through the C ABI): the reference's own five
known answers (test/lifelong_metrics_test.cpp:33-176) and random candidate sets against the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import lifelong

pytestmark = pytest.mark.gpu


def test_reference_known_answers(kartohip_lib):
    from slam_toolbox_amd.lifelong import computeScores
    s1 = lifelong.ScanBox(barycenter=(3.5, 4.0), bbox_size=(3.0, 4.0), unique_id=50)
    s2 = lifelong.ScanBox(barycenter=(3.5, 5.5), bbox_size=(3.0, 3.0), points=np.array([[3.0, 5.0], [3.0, 3.0]]),
                          unique_id=5, n_edges=2)
    kept, iou, area, reading, score = computeScores(s1, [s2])
    assert kept[0]
    assert iou[0] == 0.4                                  # TestIntersectOverUnion
    assert abs(area[0] - 0.6666) <= 0.01                  # TestAreaOverlap
    assert reading[0] == 0.5                              # TestPtOverlap


def test_random_candidate_sets_against_the_oracle(kartohip_lib):
    from slam_toolbox_amd.lifelong import computeScores
    rng = np.random.default_rng(31)
    for trial in range(6):
        ref = lifelong.ScanBox(barycenter=tuple(rng.uniform(0, 10, 2)), bbox_size=tuple(rng.uniform(4, 12, 2)),
                               unique_id=400 + trial)
        cands = []
        for k in range(300):
            centre = np.asarray(ref.barycenter) + rng.normal(0, 3.0, 2)
            n_pts = int(rng.integers(1, 1081))
            cands.append(lifelong.ScanBox(barycenter=tuple(centre), bbox_size=tuple(rng.uniform(2, 14, 2)),
                                          points=centre + rng.normal(0, 3.0, (n_pts, 2)),
                                          unique_id=int(rng.integers(0, 420)), n_edges=int(rng.integers(1, 7)),
                                          score=float(rng.uniform(0.1, 1.0))))
        cands[0].unique_id = 0                                     # the critical lynch-points keep their score
        cands[1].unique_id = 1
        cands[2].barycenter, cands[2].bbox_size, cands[2].n_edges, cands[2].unique_id = ref.barycenter, ref.bbox_size, 2, 100   # IoU 1, old node: -1
        p = lifelong.DecayParams(scan_buffer_size=10 + trial)
        kept, iou, area, reading, score = computeScores(ref, cands, p)
        o_kept, o_iou, o_area, o_reading, o_score = lifelong.compute_scores(ref, cands, p)
        assert np.array_equal(kept, o_kept)
        for got, want in ((iou, o_iou), (area, o_area), (reading, o_reading), (score, o_score)):
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
        assert score[2] == -1.0 and kept.sum() > 20
