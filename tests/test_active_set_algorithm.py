"""CPU: the order-dependent rule of AddScan -- "skip the point if its cell is already 100" (Mapper.cpp:1093-1096), which bites
when the smear kernel holds 100 off-centre (sigma / resolution >= 9.99: the centre's 4-neighbours, the sequential preset of
offline.yaml) -- in the two forms the code base uses:

  sequential  points in order; a stamped point turns its cell and the footprint cells 100; a later point landing on a
              100-cell is skipped (the reference, and oracle/karto_oracle.c)
  parallel    K1a on the device (csrc/matcher_kernels.hip: k_cell_first / k_cell_links / k_active_set): per cell only its
              FIRST point can stamp; a cell is active iff no footprint neighbour whose first point comes earlier is
              active -- the lexicographically first maximal independent set, reached by sweeps in any order

The kernels are covered by the GPU parity tests (preset S everywhere); this pins the equivalence of the two forms."""
import numpy as np
import pytest

FOOT = [(1, 0), (-1, 0), (0, 1), (0, -1)]


def sequential(cells):
    hundred = set()
    active = np.zeros(len(cells), dtype=bool)
    for p, c in enumerate(cells):
        if c in hundred:
            continue
        active[p] = True
        hundred.add(c)
        for dx, dy in FOOT:
            hundred.add((c[0] + dx, c[1] + dy))
    return active


def parallel(cells, rng):
    first = {}
    for p, c in enumerate(cells):                       # k_cell_first: atomicMin on the point index
        if c not in first:
            first[c] = p
    keys = list(first)
    state = {c: 0 for c in keys}                        # 0 undecided, 1 active, 2 inactive
    nbrs = {c: [(c[0] + dx, c[1] + dy) for dx, dy in FOOT if (c[0] + dx, c[1] + dy) in first and first[(c[0] + dx, c[1] + dy)] < first[c]]
            for c in keys}                              # k_cell_links: occupied neighbours whose first point is earlier
    sweeps = 0
    while any(s == 0 for s in state.values()):
        sweeps += 1
        order = list(keys)
        rng.shuffle(order)                              # the sweeps of k_active_set see the cells in no particular order
        for c in order:
            if state[c]:
                continue
            st = [state[n] for n in nbrs[c]]
            if any(s == 1 for s in st):
                state[c] = 2
            elif all(s == 2 for s in st):
                state[c] = 1
        assert sweeps < 10000
    active = np.zeros(len(cells), dtype=bool)
    for c, p in first.items():
        active[p] = state[c] == 1
    return active, sweeps


def _walls(rng, n):
    """scan-like point sequences: walls sampled densely (neighbouring readings share or touch cells), revisited by later scans"""
    cells = []
    for _ in range(n):
        x0, y0 = rng.integers(0, 60, 2)
        ang = rng.uniform(0, 2 * np.pi)
        length = rng.integers(5, 60)
        step = rng.uniform(0.3, 1.4)
        for k in range(length):
            cells.append((int(round(x0 + k * step * np.cos(ang))), int(round(y0 + k * step * np.sin(ang)))))
    return cells


@pytest.mark.parametrize("seed", range(6))
def test_sweeps_reach_the_sequential_result(seed):
    rng = np.random.default_rng(seed)
    cells = _walls(rng, 40)
    want = sequential(cells)
    got, sweeps = parallel(cells, rng)
    assert np.array_equal(got, want), (int(want.sum()), int(got.sum()))
    assert want.sum() > 0 and (~want).sum() > 0 and sweeps >= 1


def test_a_dense_block_and_a_long_chain():
    # a filled block (every cell has all four neighbours) and one long diagonal staircase visited in order: the dependency
    # chain is as long as the staircase
    rng = np.random.default_rng(1)
    block = [(x, y) for y in range(12) for x in range(12)]
    stairs = []
    for k in range(80):
        stairs += [(100 + k, 100 + k), (101 + k, 100 + k)]
    for cells in (block, stairs, block[::-1] + stairs[::-1]):
        want = sequential(cells)
        got, _ = parallel(cells, rng)
        assert np.array_equal(got, want)
