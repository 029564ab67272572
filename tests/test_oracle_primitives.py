"""Canonical primitives of the oracle against LAPACK/numpy (tolerance) and against their defining
identities (exact where the identity is exact).  CPU only."""
import numpy as np
import scipy.linalg as sla


def test_llt_matches_lapack(oracle):
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 31, 32, 50, 130):
        A = rng.standard_normal((n, n))
        H = A @ A.T + 1e-3 * np.eye(n)
        ret, Lf = oracle.llt(H)
        assert ret == -1
        L = np.tril(Lf)
        assert np.allclose(L, sla.cholesky(H, lower=True), rtol=1e-10, atol=1e-12)
        # strictly upper part untouched (GoldfarbIdnaniSolver.h: "G overwritten by its Cholesky factor", lower only)
        assert np.array_equal(np.triu(Lf, 1), np.triu(H, 1))


def test_llt_failure_index(oracle):
    H = np.diag([1.0, 2.0, -1.0, 3.0])
    ret, Lf = oracle.llt(H)
    assert ret == 2
    assert Lf[3, 3] == 3.0  # nothing written at/after the failing column
    ret, _ = oracle.llt(np.array([[0.0]]))
    assert ret == 0


def test_inv_lt(oracle):
    rng = np.random.default_rng(4)
    for n in (1, 3, 10, 50):
        A = rng.standard_normal((n, n))
        ret, Lf = oracle.llt(A @ A.T + np.eye(n))
        J = oracle.inv_lt(Lf)
        L = np.tril(Lf)
        assert np.array_equal(np.tril(J, -1), np.zeros((n, n)))  # exact zeros below the diagonal
        assert np.allclose(L.T @ J, np.eye(n), atol=1e-10)


def test_givens_convention(oracle):
    """Eigen JacobiRotation::makeGivens real case: [c -s; s c] [p; q] = [r; 0], r >= 0."""
    rng = np.random.default_rng(5)
    for p, q in list(rng.standard_normal((200, 2))) + [(3.0, 4.0), (-3.0, 4.0), (4.0, -3.0), (1.0, 1.0), (-1.0, -1.0)]:
        c, s, r = oracle.givens(p, q)
        assert r >= 0 and abs(r - np.hypot(p, q)) <= 4e-16 * r
        assert abs(c * p - s * q - r) <= 1e-15 * max(r, 1)
        assert abs(s * p + c * q) <= 1e-15 * max(r, 1)
    assert oracle.givens(2.0, 0.0) == (1.0, 0.0, 2.0)
    assert oracle.givens(-2.0, 0.0) == (-1.0, 0.0, 2.0)
    assert oracle.givens(0.0, 2.0) == (0.0, -1.0, 2.0)
    assert oracle.givens(0.0, -2.0) == (0.0, 1.0, 2.0)
    assert oracle.givens(0.0, 0.0) == (1.0, 0.0, 0.0)


def test_dot_is_sequential_fma_chain(oracle):
    import math

    rng = np.random.default_rng(6)
    a = rng.standard_normal(37)
    b = rng.standard_normal(37)
    acc = 0.0
    try:
        fma = math.fma  # python >= 3.13
    except AttributeError:
        from fractions import Fraction

        def fma(x, y, z):
            return float(Fraction(x) * Fraction(y) + Fraction(z))  # exact rational, rounded once

    for k in range(37):
        acc = fma(a[k], b[k], acc)
    assert oracle.dot(a, b) == acc


def test_dot_blk_is_segments_of_64_and_a_pairwise_tree(oracle):
    """CANON C8 (oracle/jrlqp_oracle.c orc_dot_blk): up to 256 entries the C1 chain; beyond, per segment [64 s, 64 s + 63] an ascending
    fma chain from +0 over the entries >= lo, the segment sums combined by the adjacent-first pairwise tree."""
    from fractions import Fraction

    def fma(x, y, z):
        return float(Fraction(x) * Fraction(y) + Fraction(z))  # exact rational, rounded once

    rng = np.random.default_rng(8)
    for n, lo in ((200, 0), (256, 7), (257, 0), (300, 0), (1000, 0), (1000, 130), (1024, 64), (1500, 1499), (4096, 0)):
        a = rng.standard_normal(n) * np.exp(rng.uniform(-8, 8, n))
        b = rng.standard_normal(n)
        if n <= 256:
            acc = 0.0
            for k in range(lo, n):
                acc = fma(a[k], b[k], acc)
            want = acc
        else:
            p = [0.0] * 64
            for s in range((n + 63) // 64):
                acc = 0.0
                for k in range(max(64 * s, lo), min(64 * s + 64, n)):
                    acc = fma(a[k], b[k], acc)
                p[s] = acc
            h = 1
            while h < 64:
                for i in range(0, 64, 2 * h):
                    p[i] = p[i] + p[i + h]
                h *= 2
            want = p[0]
        assert oracle.dot_blk(a, b, lo) == want, (n, lo)
    assert oracle.dot_blk(np.ones(10), np.ones(10)) == oracle.dot(np.ones(10), np.ones(10)) == 10.0
