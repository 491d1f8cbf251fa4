"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement of the lifelong-mapping node-decay scoring of slam_toolbox:
src/experimental/slam_toolbox_lifelong.cpp:199-250 (computeObjectiveScore), :253-292 (computeScore),
:295-329 (computeScores), :373-478 (computeIntersectBounds / computeIntersect / computeIntersectOverUnion /
computeAreaOverlapRatio / computeReadingOverlapRatio); parameter defaults :60-100.
PINNED with the reference's own known answers, the five cases of test/lifelong_metrics_test.cpp:33-176
(tests/test_lifelong_oracle.py).  The file itself needs rclcpp and cannot be compiled here."""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class DecayParams:
    iou_thresh: float = 0.10           # lifelong_minimum_score
    iou_match: float = 0.85            # lifelong_iou_match
    removal_score: float = 0.10        # lifelong_node_removal_score
    overlap_scale: float = 0.5         # lifelong_overlap_score_scale
    constraint_scale: float = 0.05     # lifelong_constraint_multiplier
    nearby_penalty: float = 0.001      # lifelong_nearby_penalty
    candidates_scale: float = 0.03     # lifelong_candidates_scale
    scan_buffer_size: int = 10         # mapper scan_buffer_size (offline.yaml:35)


@dataclass
class ScanBox:
    """What the scoring reads from a LocalizedRangeScan / Vertex: barycenter pose, bounding-box size,
    filtered point readings (GetPointReadings(true)), unique id, edge count, current vertex score."""
    barycenter: tuple
    bbox_size: tuple
    points: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    unique_id: int = 0
    n_edges: int = 0
    score: float = 1.0


def intersect_bounds(s1, s2):
    up = lambda s, k: s.barycenter[k] + (s.bbox_size[k] / 2.0)      # noqa: E731
    lo = lambda s, k: s.barycenter[k] - (s.bbox_size[k] / 2.0)      # noqa: E731
    x_u, y_u = min(up(s1, 0), up(s2, 0)), min(up(s1, 1), up(s2, 1))
    x_l, y_l = max(lo(s1, 0), lo(s2, 0)), max(lo(s1, 1), lo(s2, 1))
    return x_l, x_u, y_l, y_u


def intersect(s1, s2):
    x_l, x_u, y_l, y_u = intersect_bounds(s1, s2)
    v = (y_u - y_l) * (x_u - x_l)
    return 0.0 if v < 0.0 else v


def intersect_over_union(s1, s2):
    i = intersect(s1, s2)
    uni = (s1.bbox_size[0] * s1.bbox_size[1]) + (s2.bbox_size[0] * s2.bbox_size[1]) - i
    return i / uni


def area_overlap_ratio(ref, cand):
    return intersect(ref, cand) / (cand.bbox_size[1] * cand.bbox_size[0])


def reading_overlap_ratio(ref, cand):
    x_l, x_u, y_l, y_u = intersect_bounds(ref, cand)
    pts = np.asarray(cand.points, dtype=np.float64).reshape(-1, 2)
    inner = int(np.sum((pts[:, 0] < x_u) & (pts[:, 0] > x_l) & (pts[:, 1] < y_u) & (pts[:, 1] > y_l)))
    return float(inner) / float(pts.shape[0]) if pts.shape[0] else float("nan")     # 0/0 in the reference too


def objective_score(iou, area_overlap, reading_overlap, num_constraints, initial_score, num_candidates, p):
    if iou > p.iou_match and num_constraints < 3:
        return -1.0
    overlap = p.overlap_scale * min(area_overlap, reading_overlap)
    csf = min(1.0, max(0.0, p.constraint_scale * (num_constraints - 2)))
    csf = min(csf, overlap)
    score = initial_score * (1.0 + csf) - overlap - p.nearby_penalty
    return 1.0 if score > 1.0 else score


def compute_scores(reference, candidates, p=None):
    """computeScores: -> (kept mask, iou, area, reading, score) arrays over the candidates; candidates failing
    the IoU / edge-count filter are dropped (kept False) and do not count as candidates."""
    p = p or DecayParams()
    n = len(candidates)
    iou = np.array([intersect_over_union(reference, c) for c in candidates])
    kept = np.array([not (iou[k] < p.iou_thresh or candidates[k].n_edges < 2) for k in range(n)], dtype=bool)
    num = int(kept.sum())
    area = np.array([area_overlap_ratio(reference, c) for c in candidates])
    reading = np.array([reading_overlap_ratio(reference, c) for c in candidates])
    score = np.zeros(n)
    for k, c in enumerate(candidates):
        if not kept[k]:
            continue
        lynch = c.unique_id in (0, 1)
        if reference.unique_id - c.unique_id < p.scan_buffer_size or lynch:
            score[k] = c.score
        else:
            score[k] = objective_score(iou[k], area[k], reading[k], c.n_edges, c.score, num, p)
    return kept, iou, area, reading, score


def evaluate_node_depreciation(reference_id, boxes, adjacency, reference_xy, p=None):
    """LifelongSlamToolbox::evaluateNodeDepreciation (slam_toolbox_lifelong.cpp:149-178) for the scan just added, with
    lifelong_search_use_tree false: radius = half the diagonal of the scan's bounding box (:158-160),
    FindNearLinkedVertices = the breadth-first walk over the graph's edges that stops at vertices farther than the radius
    from the scan (Mapper.cpp:1263-1333, oracle/loops.py), computeScores (:295-329: the IoU / edge-count filter ERASES
    candidates in place, the survivors are scored with the surviving count), then in that order: score below
    lifelong_node_removal_score -> removeFromSlamGraph, else updateScoresSlamGraph.

    boxes: scan id -> ScanBox of every scan still in the map (n_edges = GetEdges().size(), score = Vertex::GetScore());
    adjacency: scan id -> adjacent scan ids in GetAdjacentVertices order; reference_xy: scan id -> GetReferencePose() xy.
    Returns the decisions in the reference's order: [("remove", id) | ("score", id, value)]."""
    from collections import deque
    p = p or DecayParams()
    ref = boxes[reference_id]
    w, h = ref.bbox_size
    radius = np.sqrt(w * w + h * h) / 2.0
    lim = radius * radius - 1e-06                              # NearScanVisitor::Visit, Mapper.cpp:1326-1327 (KT_TOLERANCE)
    centre = reference_xy[reference_id]
    near, seen, todo = [], {reference_id}, deque([reference_id])
    while todo:
        v = todo.popleft()
        dx, dy = reference_xy[v][0] - centre[0], reference_xy[v][1] - centre[1]
        if dx * dx + dy * dy <= lim:
            near.append(v)
            for u in adjacency[v]:
                u = int(u)
                if u not in seen:
                    seen.add(u)
                    todo.append(u)
    cands = [boxes[v] for v in near]
    kept, _, _, _, score = compute_scores(ref, cands, p)
    out = []
    for k, v in enumerate(near):
        if not kept[k]:
            continue
        if score[k] < p.removal_score:
            out.append(("remove", v))
        else:
            out.append(("score", v, float(score[k])))
    return out
