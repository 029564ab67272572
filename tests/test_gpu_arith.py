"""GPU: the arithmetic contract the bit-exact parity rests on -- device fp64 /, sqrt, fma and the Givens generator
are bit-identical to the host's IEEE-754 results; v_mfma_f64_16x16x4_f64 accumulates as a k-ordered fma chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(rng, n):
    # wide dynamic range, both signs, plus special magnitudes
    x = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
    x[:8] = [1.0, -1.0, 3.0, 1e-300, 1e300, 2.0**-1060, 0.1, 7.0]
    return x


def test_div_sqrt_fma_givens_bit_exact(pkg, oracle):
    import torch
    import jrl_qp_amd.capi as capi

    rng = np.random.default_rng(2024)
    n = 1 << 20
    a, b, c = _rand(rng, n), _rand(rng, n), _rand(rng, n)
    da, db, dc = (torch.from_numpy(v).cuda() for v in (a, b, c))
    ddiv, dsqrt, dfma, gc, gs, gr = (t.cpu().numpy() for t in capi.selftest_arith(da, db, dc))
    with np.errstate(all="ignore"):
        assert np.array_equal(ddiv, a / b)
        assert np.array_equal(dsqrt, np.sqrt(np.abs(a)))
    # the lean Givens generator of the removal kernels (make_givens_lean) is compared with make_givens on the device, operand by
    # operand: a difference comes back as NaN
    assert np.isfinite(gc).all() and np.isfinite(gs).all() and np.isfinite(gr).all(), "make_givens_lean differs from make_givens"
    # fma and givens against the oracle's libm/hardware versions on a sample
    idx = rng.integers(0, n, 20000)
    import ctypes as ct

    for i in idx[:20000]:
        cc, ss, rr = oracle.givens(a[i], b[i])
        assert (gc[i], gs[i], gr[i]) == (cc, ss, rr), (i, a[i], b[i])
    import math

    if hasattr(math, "fma"):
        ref = np.array([math.fma(a[i], b[i], c[i]) for i in idx])
    else:
        # exact rational evaluation, rounded once
        from fractions import Fraction

        ref = np.array([float(Fraction(a[i]) * Fraction(b[i]) + Fraction(c[i])) for i in idx[:2000]])
        idx = idx[:2000]
    assert np.array_equal(dfma[idx], ref)


def test_mfma_f64_is_k_ordered_fma_chain(pkg):
    """Pins how one v_mfma_f64_16x16x4_f64 rounds: D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0,C)))).
    (Needed by the blocked-LLT trailing update of the large-QP path to stay bit-identical to the oracle.)"""
    import torch
    import jrl_qp_amd.capi as capi
    from fractions import Fraction

    rng = np.random.default_rng(7)
    A = rng.standard_normal((16, 4)) * np.exp(rng.uniform(-8, 8, (16, 4)))
    B = rng.standard_normal((4, 16)) * np.exp(rng.uniform(-8, 8, (4, 16)))
    C = rng.standard_normal((16, 16))
    D = capi.selftest_mfma_f64(*(torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in (A, B, C))).cpu().numpy()

    def fma(x, y, z):
        return float(Fraction(x) * Fraction(y) + Fraction(z))

    ref = np.empty((16, 16))
    for i in range(16):
        for j in range(16):
            acc = C[i, j]
            for k in range(4):
                acc = fma(A[i, k], B[k, j], acc)
            ref[i, j] = acc
    assert np.allclose(D, A @ B + C, rtol=1e-12, atol=1e-12)  # layout sanity first
    nbad = int(np.sum(D != ref))
    assert nbad == 0, f"{nbad}/256 elements differ from the k-ordered fma chain"
