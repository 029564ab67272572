"""The Givens coefficients of an addition (CANON C7, oracle orc_givens_chain): radii from a suffix sum of squares taken in a
fixed parallel order, every (c_i, s_i) one division by its radius; entries of d that cannot be squared (beyond 2^+-500, not
finite) send the whole chain through the serial makeGivens recurrence instead.  The branch is a pure function of d, so the
kernels and the oracle must take it together.  These tests drive both branches through the three kernels that own a copy of
the chain (gi_batch_kernel.h, gi_batch_big.hip, gi_dense.hip add_chain_kernel) and require BIT-identical results:
  * ordinary problems never need the recurrence (the phase-profile build counts the fallbacks: zero);
  * problems whose Hessian couples the variables through entries of 1e-170 -- rows of J = L^-T, i.e. d for a bound constraint,
    then hold non-zero entries far below 2^-500 -- take the recurrence, in the kernels as in the oracle."""
import numpy as np
import pytest

from test_gpu_batch import assert_bit_identical, run_gpu
from test_gpu_dense import check, solve_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


def tiny_coupling_batch(P, B, n, m, seed):
    """min 1/2 x^T (I + 1e-170 S) x + a.x  s.t.  -0.2 <= x <= 0.2 (and a few slack general constraints): every bound that activates has
    d = a row of J = L^-T, whose off-diagonal entries are ~1e-170 -- representable, non-zero, and not squarable."""
    rng = np.random.default_rng(seed)
    S = rng.standard_normal((B, n, n))
    S = S + np.transpose(S, (0, 2, 1))
    G = np.broadcast_to(np.eye(n), (B, n, n)) + 1e-170 * S
    a = rng.standard_normal((B, n))
    C = rng.standard_normal((B, m, n))
    bl, bu = np.full((B, m), -50.0), np.full((B, m), 50.0)
    xl, xu = np.full((B, n), -0.2), np.full((B, n), 0.2)
    z = np.zeros
    return P.PackedBatch(n=n, m=m, nb=n, G=np.ascontiguousarray(G), a=a, C=C, bl=bl, bu=bu, xl=xl, xu=xu, x_star=z((B, n)), lam_star=z((B, m + n)))


def fallbacks_and_additions(capi, pb, use_bounds):
    import torch

    prof = torch.zeros((pb.B, 32), dtype=torch.int64, device="cuda")
    capi.set_profile(capi.PROFILE_GI_BATCH, prof)
    try:
        run_gpu(capi, pb, use_bounds=use_bounds)
    finally:
        capi.set_profile(capi.PROFILE_GI_BATCH, None)
    pr = prof.cpu().numpy()
    return int(pr[:, 13].sum()), int(pr[:, 10].sum())


BATCH_CASES = [
    dict(nVar=50, nObj=50, nIneq=200, nStrongActIneq=15),  # register-resident C, role-specialised waves
    dict(nVar=64, nObj=64, nIneq=128, nStrongActIneq=19),
    dict(nVar=30, nObj=30, nEq=3, nIneq=60, nStrongActIneq=8, bounds=True, nStrongActBounds=4),  # generic kernel, bounds: zero tails of d
    dict(nVar=100, nObj=100, nIneq=150, nStrongActIneq=20),  # gi_batch_big.hip: two blocks of the scan
    dict(nVar=200, nObj=200, nIneq=100, nStrongActIneq=30),  # four blocks
]


@pytest.mark.parametrize("ci", range(len(BATCH_CASES)))
def test_ordinary_problems_use_the_tree(capi, P, oracle, ci):
    ch = P.Characteristics(**BATCH_CASES[ci])
    pb = P.generate_batch(96 if ch.nVar <= 64 else 24, ch, first_index=31000 + 1000 * ci)
    hb = ch.bounds
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None, nthreads=8)
    gpu = run_gpu(capi, pb, use_bounds=hb)
    assert_bit_identical(gpu, ref, f"case {ci}")
    if ci < 2:  # (the kernels whose phase-profile build counts the fallbacks in slot 13)
        fb, adds = fallbacks_and_additions(capi, pb, hb)
        assert adds > 0 and fb == 0, (fb, adds)


@pytest.mark.parametrize("n,m", [(30, 4), (50, 8), (64, 0), (100, 6), (160, 4)])
def test_unsquarable_entries_take_the_recurrence(capi, P, oracle, n, m):
    pb = tiny_coupling_batch(P, 48 if n <= 64 else 12, n, m, seed=4200 + n)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl, pb.xu, nthreads=8)
    assert (ref["term"] == 0).all() and ref["iters"].min() >= 2
    gpu = run_gpu(capi, pb, use_bounds=True)
    assert_bit_identical(gpu, ref, f"tiny coupling n={n}")  # (a tree over these entries would underflow to zero radii: NaN coefficients)


def test_oracle_branches(oracle):
    """The oracle's own switch: in range -> tree (returns 1), one entry out of range or not finite -> recurrence (returns 0); both give
    c^2 + s^2 = 1 to rounding and the same radius to a few ulps."""
    import ctypes as ct

    L = oracle.lib
    c_dp = ct.POINTER(ct.c_double)
    L.orc_givens_chain.restype = ct.c_int
    L.orc_givens_chain.argtypes = [ct.c_int, ct.c_int, c_dp, c_dp, c_dp, c_dp]
    rng = np.random.default_rng(5)
    n, q = 150, 7

    def run(d):
        c, s, r = np.zeros(n), np.zeros(n), ct.c_double()
        t = L.orc_givens_chain(n, q, d.ctypes.data_as(c_dp), c.ctypes.data_as(c_dp), s.ctypes.data_as(c_dp), ct.byref(r))
        return t, c[q:n - 1], s[q:n - 1], r.value

    d = rng.standard_normal(n)
    t, c, s, r = run(d)
    assert t == 1 and np.abs(c * c + s * s - 1).max() < 1e-15 and abs(r - np.linalg.norm(d[q:])) < 1e-13
    for bad in (1e-170, 1e170, np.inf, np.nan):
        d2 = d.copy()
        d2[40] = bad
        assert run(d2)[0] == 0
    d3 = d.copy()
    d3[40] = 1e-170
    t3, c3, s3, r3 = run(d3)
    assert t3 == 0 and np.abs(c3 * c3 + s3 * s3 - 1).max() < 1e-15 and abs(r3 / np.linalg.norm(d3[q:]) - 1) < 1e-14
    d4 = d.copy()
    d4[60:] = 0.0  # a zero tail: rotations over it are the identity
    t4, c4, s4, r4 = run(d4)
    assert t4 == 1 and np.all(c4[60 - q:] == 1.0) and np.all(s4[60 - q:] == 0.0)


@pytest.mark.parametrize("tiny", [False, True])
def test_dense_path(capi, P, oracle, tiny):
    if tiny:
        pb = tiny_coupling_batch(P, 1, 200, 10, seed=99)
        ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], pb.xl[0], pb.xu[0], max_iter=3000)
    else:
        pb = P.generate_batch(1, P.benchmark_characteristics(200, 300), first_index=5151)
        ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=3000)
    assert ref.term == 0 and ref.iters >= 2
    gpu = solve_gpu(capi, pb, 0, tiny, max_iter=3000)
    check(gpu, ref, f"dense path, tiny coupling {tiny}")
