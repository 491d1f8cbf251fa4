"""CPU: the round-3 level pipeline of the solver (csrc/spa_kernels.hip: k_potrf / k_trsm / k_syrk / k_backward3) restated
in numpy with the kernels' own index rules, against plain dense linear algebra:

  * k_potrf factors the ns x ns pivot block in an nsp x nsp matrix (nsp = ns rounded up to 16, identity on the padding) and
    computes W = L^-T in the SAME sweep: the rows of the identity ride along as extra rows of the panel (x L^T = e_i) and
    live in the strictly upper triangle of the matrix (the diagonal of W = the reciprocal pivots, kept apart); rows of the
    identity that start inside the current panel go through a 16 x 16 side buffer (Xd);
  * k_trsm multiplies a slab of F21 with W block column by block column, K <= J only (W is block upper triangular and the
    diagonal blocks carry explicit zeros below the diagonal);
  * forward: y1 = W^T b1, b2 -= L21 y1; backward: x1 = W (y1 - L21^T x2).

The kernels are covered on the GPU by tests/test_spa_gpu.py (solver parity, residual of every linear solve, agreement
with the panel-pair kernels); this file pins the formulation for every tail shape of ns without a GPU."""
import numpy as np
import pytest

NB = 16


def potrf_with_inverse(F11):
    """returns (A, rd): A lower = L, A strictly upper = strictly upper part of L^-T, rd = 1 / diag(L)"""
    ns = F11.shape[0]
    nsp = (ns + NB - 1) // NB * NB
    nt = nsp // NB
    A = np.zeros((nsp, nsp))
    A[:ns, :ns] = np.tril(F11)
    for i in range(ns, nsp):
        A[i, i] = 1.0
    rd = np.zeros(nsp)
    Xd = np.zeros((NB, NB))

    def diag(jb):
        c0 = NB * jb
        blk = A[c0:c0 + NB, c0:c0 + NB]
        L = np.linalg.cholesky(np.tril(blk) + np.tril(blk, -1).T)
        for r in range(NB):
            for c in range(r + 1):
                A[c0 + r, c0 + c] = L[r, c]           # lower part only: the upper part belongs to W
        rd[c0:c0 + NB] = 1.0 / np.diag(L)
        return L

    def row_solve(x, L):                              # x L^T = a, column by column
        x = x.copy()
        for c in range(NB):
            x[c] = x[c] / L[c, c]
            x[c + 1:] -= x[c] * L[c + 1:, c]
        return x

    L = diag(0)
    for jb in range(nt):
        c0 = NB * jb
        for i in range(nsp):
            own = c0 <= i < c0 + NB
            a = np.zeros(NB)
            if own:
                a[i - c0] = 1.0
            else:
                a = A[i, c0:c0 + NB].copy()
            x = row_solve(a, L)
            if own:
                Xd[i - c0, :] = x
                for c in range(NB):
                    if c > i - c0:
                        A[i, c0 + c] = x[c]
            else:
                A[i, c0:c0 + NB] = x
        if jb + 1 >= nt:
            break
        # tiles: (I, J) with I >= J > jb (pivot block), (I, J) with I <= jb < J (identity rows)
        new = A.copy()
        for bi in range(jb + 1, nt):
            for bj in range(jb + 1, bi + 1):
                B = A[NB * bi:NB * bi + NB, c0:c0 + NB]
                Aj = A[NB * bj:NB * bj + NB, c0:c0 + NB]
                upd = B @ Aj.T
                for r in range(NB):
                    for c in range(NB):
                        if bi != bj or c <= r:
                            new[NB * bi + r, NB * bj + c] -= upd[r, c]
        for ie in range(jb + 1):
            for bj in range(jb + 1, nt):
                B = Xd if ie == jb else A[NB * ie:NB * ie + NB, c0:c0 + NB]
                Aj = A[NB * bj:NB * bj + NB, c0:c0 + NB]
                new[NB * ie:NB * ie + NB, NB * bj:NB * bj + NB] -= B @ Aj.T
        A = new
        L = diag(jb + 1)
    return A, rd


def w_from(A, rd):
    """W (nsp x nsp, column-major L^-1 = row-major L^-T with zeros left of the diagonal) as k_potrf stores it"""
    nsp = A.shape[0]
    Wt = np.triu(A, 1) + np.diag(rd)          # Wt[q][j] = (L^-T)[q][j]
    return Wt


@pytest.mark.parametrize("ns", [3, 9, 15, 16, 17, 30, 32, 33, 36, 47, 48, 63, 64, 66, 81, 96, 111, 126, 128])
def test_potrf_with_riding_identity(ns):
    rng = np.random.default_rng(ns)
    G = rng.normal(size=(ns, ns + 5))
    F11 = G @ G.T + ns * np.eye(ns)
    A, rd = potrf_with_inverse(F11)
    L = np.linalg.cholesky(F11)
    assert np.allclose(np.tril(A)[:ns, :ns], L, rtol=1e-12, atol=1e-12)
    Wt = w_from(A, rd)
    assert np.allclose(Wt[:ns, :ns], np.linalg.inv(L).T, rtol=1e-10, atol=1e-12)
    # padding: identity
    assert np.allclose(Wt[ns:, ns:], np.eye(A.shape[0] - ns)) and np.allclose(Wt[:ns, ns:], 0.0)


@pytest.mark.parametrize("ns,nu", [(3, 6), (36, 171), (48, 33), (81, 150), (126, 294)])
def test_trsm_syrk_and_solves(ns, nu):
    rng = np.random.default_rng(ns + nu)
    m = ns + nu
    G = rng.normal(size=(m, m + 3))
    F = G @ G.T + m * np.eye(m)
    A, rd = potrf_with_inverse(F[:ns, :ns])
    nsp = A.shape[0]
    nt = nsp // NB
    Wt = w_from(A, rd)
    # k_trsm: X[r][16 J + j] = sum_{K <= J} S[r][K block] . (L^-T)[K block][J block]
    S = np.zeros((nu, nsp))
    S[:, :ns] = F[ns:, :ns]
    X = np.zeros((nu, nsp))
    for J in range(nt):
        for K in range(J + 1):
            X[:, NB * J:NB * J + NB] += S[:, NB * K:NB * K + NB] @ Wt[NB * K:NB * K + NB, NB * J:NB * J + NB]
    L = np.linalg.cholesky(F)
    assert np.allclose(X[:, :ns], L[ns:, :ns], rtol=1e-10, atol=1e-11)
    assert np.allclose(X[:, ns:], 0.0)
    # k_syrk
    F22 = F[ns:, ns:] - X @ X.T
    assert np.allclose(np.tril(F22), np.tril(L[ns:, ns:] @ L[ns:, ns:].T), rtol=1e-9, atol=1e-9)
    # forward / backward with W against a dense solve of the whole front system
    b = rng.normal(size=m)
    y1 = Wt[:ns, :ns].T @ b[:ns]
    b2 = b[ns:] - X[:, :ns] @ y1
    y2 = np.linalg.solve(L[ns:, ns:], b2)
    x2 = np.linalg.solve(L[ns:, ns:].T, y2)
    x1 = Wt[:ns, :ns] @ (y1 - X[:, :ns].T @ x2)
    assert np.allclose(np.concatenate([x1, x2]), np.linalg.solve(F, b), rtol=1e-9, atol=1e-10)
