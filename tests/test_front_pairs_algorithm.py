"""CPU: the panel-pair partial Cholesky of a front as k_factor2 runs it (csrc/spa_kernels.hip) -- pairs of 16-column panels
kept in one 32-column panel P, thin update of the second panel inside P, ONE trailing update per pair, and the tail rules:
a last pair whose second panel has fewer than 16 pivots, a last single panel with fewer than 16, non-pivot columns never
loaded into P, the trailing update starting at the first 16-aligned row outside a FULL panel (R0) with the pivot columns
masked -- restated in numpy against a plain partial Cholesky (L11, L21 and the Schur complement).  The kernel is covered by
the solver's GPU parity tests; this pins the index rules for every tail shape."""
import numpy as np
import pytest

NB = 16


def factor_front_in_pairs(F, ns):
    """F: symmetric positive definite front (m x m), first ns columns are pivots.  Returns the lower triangle after the
    partial factorisation: L in the pivot columns, the Schur complement behind them."""
    A = np.tril(F).copy()
    m = A.shape[0]
    jb = 0
    while jb < ns:
        rem = ns - jb
        nb_a = min(NB, rem)
        nb_b = min(NB, rem - NB) if rem > NB else 0
        npiv = nb_a + nb_b
        M = m - jb
        R0 = (NB if nb_a == NB else 0) + (NB if nb_b == NB else 0)
        P = np.zeros((M + NB, 2 * NB))
        for c in range(npiv):                                     # pivot columns only, lower part
            P[c:M, c] = A[jb + c:m, jb + c]
        # first diagonal block, rows below it
        L11 = np.linalg.cholesky(P[:nb_a, :nb_a] + np.tril(P[:nb_a, :nb_a], -1).T)
        P[:nb_a, :nb_a] = L11
        P[nb_a:M, :nb_a] = np.linalg.solve(L11, P[nb_a:M, :nb_a].T).T
        if nb_b > 0:
            # thin update: the second panel's pivot columns, rows from its diagonal block down
            upd = P[NB:M, :NB] @ P[NB:NB + nb_b, :NB].T
            P[NB:M, NB:NB + nb_b] -= upd
            blk = P[NB:NB + nb_b, NB:NB + nb_b]
            L22 = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
            P[NB:NB + nb_b, NB:NB + nb_b] = L22
            P[NB + nb_b:M, NB:NB + nb_b] = np.linalg.solve(L22, P[NB + nb_b:M, NB:NB + nb_b].T).T
        # trailing update from R0, pivot columns masked (P is zero in the non-pivot columns)
        X = P[:M, :]
        T = X @ X.T
        for row in range(R0, M):
            lo = max(R0, npiv)
            if lo <= row:
                A[jb + row, jb + lo:jb + row + 1] -= T[row, lo:row + 1]
        # L columns back to the front
        for c in range(npiv):
            A[jb + c:m, jb + c] = P[c:M, c]
        jb += npiv
    return A


def reference(F, ns):
    m = F.shape[0]
    L11 = np.linalg.cholesky(F[:ns, :ns])
    out = np.zeros_like(F)
    out[:ns, :ns] = L11
    if ns < m:
        L21 = np.linalg.solve(L11, F[ns:, :ns].T).T
        out[ns:, :ns] = L21
        out[ns:, ns:] = np.tril(F[ns:, ns:] - L21 @ L21.T)
    return out


@pytest.mark.parametrize("m,ns", [(9, 3), (40, 16), (40, 15), (40, 17), (75, 31), (75, 32), (75, 33), (120, 48), (120, 47),
                                  (120, 49), (120, 63), (120, 64), (120, 65), (78, 78), (66, 66), (33, 33), (17, 1), (200, 81)])
def test_pairs_with_tail_rules_equal_a_plain_partial_cholesky(m, ns):
    rng = np.random.default_rng(m * 1000 + ns)
    B = rng.normal(size=(m, m))
    F = B @ B.T + m * np.eye(m)
    got = factor_front_in_pairs(F, ns)
    want = reference(F, ns)
    assert np.allclose(got, want, rtol=0, atol=1e-9 * np.abs(want).max())
