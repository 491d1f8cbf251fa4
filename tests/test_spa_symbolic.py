"""CPU: the solver's symbolic analysis (csrc/spa_symbolic.cpp: nested dissection with vertex-cover separators, supernode
chains, assembly tree, level numbering) checked against a plain symbolic Cholesky of the same ordering: every front's
row structure must equal the true structure of its first column's L pattern, children must sit on lower levels with
contiguous front ids per level, and the child -> parent position maps must point at the same elimination positions."""
import os
import subprocess

import numpy as np
import pytest

from slam_toolbox_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(tmp_path, edges, n_nodes, args=()):
    exe = str(tmp_path / "sym_probe")
    if not os.path.exists(exe):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "sym_probe.cpp"),
                        os.path.join(ROOT, "slam_toolbox_amd", "csrc", "spa_symbolic.cpp"), "-o", exe], check=True, timeout=300)
    ef = str(tmp_path / "edges.bin")
    np.asarray(edges, dtype=np.int32).tofile(ef)
    dump = str(tmp_path / "sym.bin")
    out = subprocess.run([exe, ef, str(n_nodes)] + [str(a) for a in args], env=dict(os.environ, SYM_PROBE_DUMP=dump),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = np.fromfile(dump, dtype=np.int32)
    arrays, p = [], 0
    while p < raw.size:
        n = int(raw[p]); arrays.append(raw[p + 1:p + 1 + n]); p += 1 + n
    names = ["free_of_elim", "front_first", "front_ns", "front_m", "level", "parent", "rows_ptr", "rows", "child_ptr", "child_list",
             "relpos_ptr", "relpos", "cinv_ptr", "cinv"]
    assert len(arrays) == len(names)
    return dict(zip(names, arrays))


def _check(sym, edges, n_nodes):
    N = n_nodes - 1                                     # node 0 is the gauge
    perm = sym["free_of_elim"]
    assert sorted(perm.tolist()) == list(range(N)), "not a permutation"
    pos = np.empty(N, dtype=np.int64); pos[perm] = np.arange(N)
    adj = [set() for _ in range(N)]
    for a, b in edges:
        a -= 1; b -= 1
        if a >= 0 and b >= 0 and a != b:
            adj[pos[a]].add(pos[b]); adj[pos[b]].add(pos[a])
    # plain symbolic factorisation in elimination positions
    struct = [None] * N
    kids = [[] for _ in range(N)]
    for j in range(N):
        s = {w for w in adj[j] if w > j}
        for c in kids[j]:
            s |= struct[c]
        s.discard(j)
        struct[j] = s
        if s:
            kids[min(s)].append(j)
    K = len(sym["front_first"])
    first, ns, m, level, parent = sym["front_first"], sym["front_ns"], sym["front_m"], sym["level"], sym["parent"]
    covered = np.zeros(N, dtype=bool)
    padding = total_rows = 0
    for k in range(K):
        ncols = ns[k] // 3
        assert ns[k] % 3 == 0 and 1 <= ncols <= 42
        cols = range(first[k], first[k] + ncols)
        assert not covered[list(cols)].any(); covered[list(cols)] = True
        rows = sym["rows"][sym["rows_ptr"][k]:sym["rows_ptr"][k + 1]]
        assert m[k] == 3 * (ncols + len(rows))
        assert np.all(np.diff(rows) > 0) and (len(rows) == 0 or rows[0] >= first[k] + ncols)
        # a supernode's rows = the structure of its first column beyond the supernode, which must contain every later
        # column's structure (dense trapezoid); equality with the first column = no padding beyond the supernode's own rule
        true_first = {w for w in struct[first[k]] if w >= first[k] + ncols}
        union = set()
        for c in cols:
            union |= {w for w in struct[c] if w >= first[k] + ncols}
        assert union <= set(rows.tolist()), f"front {k}: fill outside its rows"
        # a supernode is treated as dense: rows beyond the true structure are explicit zeros (a separator is not always a
        # clique after its halves are gone, and a part of a split chain carries the whole chain's rows), counted below
        padding += len(rows) - len(union)
        total_rows += len(rows)
        assert true_first <= union
        # parent = front of the first row; relpos maps every row to that elimination position inside the parent
        if len(rows):
            p = parent[k]
            assert p >= 0 and first[p] <= rows[0] < first[p] + ns[p] // 3
            assert level[p] > level[k]
            prow = sym["rows"][sym["rows_ptr"][p]:sym["rows_ptr"][p + 1]]
            ppos = np.concatenate([np.arange(first[p], first[p] + ns[p] // 3), prow])
            rel = sym["relpos"][sym["relpos_ptr"][k]:sym["relpos_ptr"][k + 1]]
            assert np.array_equal(ppos[rel], rows) and np.all(np.diff(rel) > 0)
            assert k in sym["child_list"][sym["child_ptr"][p]:sym["child_ptr"][p + 1]]
        else:
            assert parent[k] == -1
    assert covered.all()
    # gather maps = inverse of relpos per (front, child)
    for k in range(K):
        mp = m[k] // 3
        kids_k = sym["child_list"][sym["child_ptr"][k]:sym["child_ptr"][k + 1]]
        assert sym["cinv_ptr"][k + 1] - sym["cinv_ptr"][k] == len(kids_k) * mp
        for s_, c in enumerate(kids_k):
            inv = sym["cinv"][sym["cinv_ptr"][k] + s_ * mp:sym["cinv_ptr"][k] + (s_ + 1) * mp]
            rel = sym["relpos"][sym["relpos_ptr"][c]:sym["relpos_ptr"][c + 1]]
            assert np.array_equal(np.nonzero(inv >= 0)[0], rel) and np.array_equal(inv[rel], np.arange(len(rel)))
    assert padding <= 0.1 * max(1, total_rows), (padding, total_rows)
    assert np.all(np.diff(level) >= 0), "front ids are not level by level"
    for l in range(level.max() + 1):
        ids = np.nonzero(level == l)[0]
        assert np.all(np.diff(m[ids]) <= 0), "fronts of a level are not sorted by size"


@pytest.mark.parametrize("n,e,seed,args", [(60, 100, 2, ()), (700, 1800, 3, ()), (3000, 9000, 4, ()), (3000, 9000, 4, (4, 5, 2)),
                                           (2000, 1999, 5, ())])
def test_symbolic_structure(tmp_path, n, e, seed, args):
    g = synth.make_pose_graph(n, e, seed=seed)
    sym = _probe(tmp_path, g["edges"], n, args)
    _check(sym, g["edges"], n)


def test_symbolic_clique_is_a_chain_of_fronts(tmp_path):
    n = 151
    edges = [(i, j) for i in range(n) for j in range(i + 1, n)]
    sym = _probe(tmp_path, edges, n)
    _check(sym, edges, n)
    assert len(sym["front_first"]) == 4 and sym["level"].tolist() == [0, 1, 2, 3]       # 150 pivots nodes -> 38 + 38 + 38 + 36
    assert sym["front_ns"].max() <= 126
