"""Structured Cholesky family of the oracle (oracle/structured_oracle.c) against the reference's own identities:
tests/blockArrowLLTTest.cpp:39-176 and tests/triBlockDiagLLTTest.cpp:35-87 (block factor/solves == dense
factor/solves to 1e-8 for every zero window), with LAPACK as the dense side; plus the stronger property the CANON
rules give: the block algorithms are BIT-identical to the oracle's dense canonical LLT / substitutions."""
import numpy as np
import pytest
import scipy.linalg as sla

BS = [3, 5, 2, 3]
OFF = np.concatenate([[0], np.cumsum(BS)])
N = int(OFF[-1])


def approx(a, b, prec=1e-8):
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b)) + 1e-300  # Eigen isApprox


def arrow_pattern(rng, first):
    B = np.zeros((N, N))
    for i, ni in enumerate(BS):
        k = OFF[i]
        B[k : k + ni, k : k + ni] = rng.uniform(-1, 1, (ni, ni))
        if first:
            B[k : k + ni, : BS[0]] = rng.uniform(-1, 1, (ni, BS[0]))
        else:
            B[k : k + ni, N - BS[-1] :] = rng.uniform(-1, 1, (ni, BS[-1]))
    B[N - BS[-1] :, N - BS[-1] :] = rng.uniform(-1, 1, (BS[-1], BS[-1]))
    return B


def dense_canonical_solve(oracle, Lfull, M, transpose):
    """Dense substitution with the same CANON order: a 1-block 'tridiagonal' system."""
    X = np.asfortranarray(M.copy())
    Lf = np.asfortranarray(Lfull.copy())
    oracle.tri_block_diag_solve([Lf], [], X, transpose=transpose)
    return X


def test_block_arrow_down(oracle):
    rng = np.random.default_rng(11)
    A = arrow_pattern(rng, False)
    H1 = np.asfortranarray(A.T @ A)
    H0 = H1.copy()
    D = [H1[OFF[i] : OFF[i + 1], OFF[i] : OFF[i + 1]] for i in range(4)]
    S = [H1[OFF[3] : OFF[4], OFF[i] : OFF[i + 1]] for i in range(3)]
    assert oracle.block_arrow_llt(D, S, up=False)
    Lref = sla.cholesky(H0, lower=True)
    assert approx(np.tril(H1), Lref)
    ret, Hd = oracle.llt(H0)
    assert ret == -1 and np.array_equal(np.tril(H1), np.tril(Hd))  # bit-identical to the dense canonical LLT
    L = np.tril(H1)
    for i in range(N):
        for j in range(i + 1, N + 1):
            B = np.zeros((N, 5), order="F")
            B[i:j] = rng.uniform(-1, 1, (j - i, 5))
            B0 = sla.solve_triangular(L, B, lower=True)
            B1 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, False, B1)
            B2 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, False, B2, i, j)
            assert approx(B1, B0) and approx(B2, B0)
            assert np.array_equal(B1, B2)  # skipping zero windows never changes a bit
            assert np.array_equal(B1, dense_canonical_solve(oracle, L, B, False))
            T0 = sla.solve_triangular(L, B, lower=True, trans="T")
            T1 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, False, T1, transpose=True)
            T2 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, False, T2, 0, j, transpose=True)
            assert approx(T1, T0) and approx(T2, T0)
            T3 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, False, T3, i, j, transpose=True)
            assert np.array_equal(T1, T3) and np.array_equal(T1, T2)
            assert np.array_equal(T1, dense_canonical_solve(oracle, L, B, True))


def test_block_arrow_up(oracle):
    rng = np.random.default_rng(12)
    A = arrow_pattern(rng, True)
    H1 = np.asfortranarray(A.T @ A)
    H3 = H1.copy()
    P = np.zeros((N, N))
    P[:3, 10:] = np.eye(3)
    P[3:, :10] = np.eye(10)
    H2 = P.T @ H1 @ P
    D = [H1[OFF[i] : OFF[i + 1], OFF[i] : OFF[i + 1]] for i in range(4)]
    S = [H1[OFF[i + 1] : OFF[i + 2], 0:3] for i in range(3)]
    assert oracle.block_arrow_llt(D, S, up=True)
    L2 = sla.cholesky(H2, lower=True)
    assert approx(np.tril(H1[3:, 3:]), L2[:10, :10])
    assert approx(np.tril(H1[:3, :3]), L2[10:, 10:])
    assert approx(H1[3:, :3].T, L2[10:, :10])
    for i in range(N):
        for j in range(i + 1, N + 1):
            B = np.zeros((N, 5), order="F")
            B[i:j] = rng.uniform(-1, 1, (j - i, 5))
            B0 = sla.solve_triangular(L2, P.T @ B, lower=True)
            B1 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, True, B1)
            B2 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, True, B2, i, j)
            assert approx(B1, B0) and approx(B2, B0) and np.array_equal(B1, B2)
            T0 = P @ sla.solve_triangular(L2, B, lower=True, trans="T")
            T1 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, True, T1, transpose=True)
            T2 = B.copy(order="F")
            oracle.block_arrow_solve(D, S, True, T2, i, j, transpose=True)
            assert approx(T1, T0) and approx(T2, T0) and np.array_equal(T1, T2)
            # both: (P L L^T P^T)^-1 B == H^-1 B
            X = B.copy(order="F")
            oracle.block_arrow_solve(D, S, True, X)
            oracle.block_arrow_solve(D, S, True, X, transpose=True)
            assert approx(X, np.linalg.solve(H3, B), 1e-7)


def test_tri_block_diag(oracle):
    rng = np.random.default_rng(13)
    Bm = np.zeros((N, N))
    for i in range(3):
        k = OFF[i]
        Bm[k : k + BS[i], k : k + BS[i]] = rng.uniform(-1, 1, (BS[i], BS[i]))
        Bm[OFF[i + 1] : OFF[i + 2], k : k + BS[i]] = rng.uniform(-1, 1, (BS[i + 1], BS[i]))
    Bm[OFF[3] :, OFF[3] :] = rng.uniform(-1, 1, (BS[3], BS[3]))
    H1 = np.asfortranarray(Bm @ Bm.T)
    H0 = H1.copy()
    D = [H1[OFF[i] : OFF[i + 1], OFF[i] : OFF[i + 1]] for i in range(4)]
    S = [H1[OFF[i + 1] : OFF[i + 2], OFF[i] : OFF[i + 1]] for i in range(3)]
    assert oracle.tri_block_diag_llt(D, S)
    assert approx(np.tril(H1), sla.cholesky(H0, lower=True))
    ret, Hd = oracle.llt(H0)
    assert ret == -1 and np.array_equal(np.tril(H1), np.tril(Hd))
    L = np.tril(H1)
    for i in range(N):
        for j in range(i + 1, N + 1):
            B = np.zeros((N, 5), order="F")
            B[i:j] = rng.uniform(-1, 1, (j - i, 5))
            B0 = sla.solve_triangular(L, B, lower=True)
            B1 = B.copy(order="F")
            oracle.tri_block_diag_solve(D, S, B1)
            B2 = B.copy(order="F")
            oracle.tri_block_diag_solve(D, S, B2, start=i)
            assert approx(B1, B0) and approx(B2, B0) and np.array_equal(B1, B2)
            assert np.array_equal(B1, dense_canonical_solve(oracle, L, B, False))
            T0 = sla.solve_triangular(L, B, lower=True, trans="T")
            T1 = B.copy(order="F")
            oracle.tri_block_diag_solve(D, S, T1, transpose=True)
            T2 = B.copy(order="F")
            oracle.tri_block_diag_solve(D, S, T2, end=j, transpose=True)
            assert approx(T1, T0) and approx(T2, T0) and np.array_equal(T1, T2)
            assert np.array_equal(T1, dense_canonical_solve(oracle, L, B, True))


def test_failure_quirks(oracle):
    """blockArrowLLT.cpp:65 / triBlockDiagLLT.cpp:19: the per-block test is `ret > 0`, so a non-positive pivot in
    column 0 of a block is NOT reported there; a later column is."""
    H = np.asfortranarray(np.diag([1.0, 2.0, 3.0, 4.0]))
    H[1, 1] = -1.0  # block 0 = diag(1,-1): fails at column 1 -> detected
    D = [H[0:2, 0:2], H[2:4, 2:4]]
    S = [H[2:4, 0:2]]
    assert not oracle.tri_block_diag_llt(D, S)
    H = np.asfortranarray(np.diag([1.0, 2.0, 3.0, 4.0]))
    H[3, 3] = -4.0
    D = [H[0:2, 0:2], H[2:4, 2:4]]
    S = [H[2:4, 0:2]]
    assert not oracle.block_arrow_llt(D, S, up=False)


def test_mpc_shaped_arrow_c5(oracle):
    """BASELINE config c5 shape (scaled down for CPU time): block-arrow(down), 20 stages x 30 + head 30."""
    rng = np.random.default_rng(14)
    nb, bs = 21, 30
    n = nb * bs
    A = np.zeros((n, n))
    for i in range(nb):
        k = i * bs
        A[k : k + bs, k : k + bs] = rng.standard_normal((bs, bs))
        A[k : k + bs, n - bs :] += rng.standard_normal((bs, bs)) * (i < nb - 1)
    H = np.asfortranarray(A.T @ A + 1e-2 * np.eye(n))
    H0 = H.copy()
    D = [H[i * bs : (i + 1) * bs, i * bs : (i + 1) * bs] for i in range(nb)]
    S = [H[n - bs :, i * bs : (i + 1) * bs] for i in range(nb - 1)]
    assert oracle.block_arrow_llt(D, S, up=False)
    assert np.allclose(np.tril(H), sla.cholesky(H0, lower=True), rtol=1e-9, atol=1e-11)
    b = np.asfortranarray(rng.standard_normal((n, 1)))
    x = b.copy(order="F")
    oracle.block_arrow_solve(D, S, False, x)
    oracle.block_arrow_solve(D, S, False, x, transpose=True)
    assert np.allclose(H0 @ x, b, atol=1e-8)
