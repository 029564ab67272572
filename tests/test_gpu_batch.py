"""GPU parity tests proper: the batched HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs -- BIT-exact on termination status, iteration count, active-set vector AND on x, u, f (the CANON rounding
contract makes the floating-point results identical, so the stated fp64 tolerance is 0 ulp)."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_gpu(capi, pb, use_bounds=False, max_iter=500, trace_cap=0, want_factor=False):
    import torch

    dev = capi.to_device(pb, "cuda:0")
    s = capi.BatchSolver(pb.n, pb.m, use_bounds, pb.B, device="cuda:0", max_iter=max_iter, trace_cap=trace_cap, want_factor=want_factor)
    r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], dev["xl"] if use_bounds else None, dev["xu"] if use_bounds else None)
    torch.cuda.synchronize()
    out = dict(x=r.x.cpu().numpy(), u=r.u.cpu().numpy(), status=r.status.cpu().numpy(), term=r.term.cpu().numpy(),
               iters=r.iters.cpu().numpy(), f=r.f.cpu().numpy())
    if want_factor:
        out["L"] = r.L.cpu().numpy()
    if trace_cap:
        out["trace"] = s.trace_numpy()
    return out


def assert_bit_identical(gpu, ref, what=""):
    for k in ("term", "iters", "status"):
        bad = np.nonzero(np.any(np.atleast_2d(gpu[k] != ref[k]).reshape(len(ref["term"]), -1), axis=1))[0]
        assert bad.size == 0, f"{what}: {k} differs for problems {bad[:10]} ({bad.size} of {len(ref['term'])})"
    for k in ("x", "u", "f"):
        same = (gpu[k] == ref[k]) | (np.isnan(gpu[k]) & np.isnan(ref[k]))
        if not same.all():
            bad = np.nonzero(~same.reshape(len(ref["term"]), -1).all(axis=1))[0]
            err = np.nanmax(np.abs(gpu[k] - ref[k]))
            raise AssertionError(f"{what}: {k} not bit-identical for problems {bad[:10]} ({bad.size}); max abs diff {err:.3e}")


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


def test_c1_small_problems(capi, P, oracle):
    """BASELINE config c1 shape (n=10, m=20, ma=3), 512 problems."""
    pb = P.generate_batch(512, P.benchmark_characteristics(10, 20))
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
    gpu = run_gpu(capi, pb, want_factor=True)
    assert_bit_identical(gpu, ref, "c1")
    assert np.all(ref["term"] == 0)
    assert np.array_equal(np.tril(gpu["L"].transpose(0, 2, 1)), np.tril(ref["L"].transpose(0, 2, 1)))  # Cholesky factor too


def test_c3_robotics_size(capi, P, oracle, pkg):
    """BASELINE config c3 shape (n=50, m=200, ma=15), 256 problems; also KKT and planted-solution checks."""
    import jrl_qp_amd.kkt as kkt

    pb = P.generate_batch(256, P.benchmark_characteristics(50, 200))
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
    gpu = run_gpu(capi, pb)
    assert_bit_identical(gpu, ref, "c3")
    assert np.all(gpu["term"] == 0)
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6
    for b in range(0, 256, 16):
        assert kkt.test_kkt(gpu["x"][b], gpu["u"][b], pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b])


@pytest.mark.parametrize("ma", [0, 25, 40])
def test_c3_other_active_fractions(capi, P, oracle, ma):
    pb = P.generate_batch(64, P.benchmark_characteristics(50, 200, ma), first_index=10000 + ma)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
    gpu = run_gpu(capi, pb)
    assert_bit_identical(gpu, ref, f"ma={ma}")


CASES = [
    dict(nVar=5, nObj=5),
    dict(nVar=5, nObj=5, nEq=2),
    dict(nVar=5, nObj=5, nIneq=8, nStrongActIneq=4),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=3),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=1, bounds=True, nStrongActBounds=2),
    dict(nVar=6, nObj=6, nIneq=10, nStrongActIneq=3, doubleSidedIneq=True),
    dict(nVar=8, nObj=8, bounds=True, nStrongActBounds=3),
    dict(nVar=10, nObj=10, nEq=3, nIneq=20, nStrongActIneq=3, bounds=True, nStrongActBounds=2, doubleSidedIneq=True),
    dict(nVar=33, nObj=33, nEq=5, nIneq=70, nStrongActIneq=10, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),
    dict(nVar=64, nObj=64, nIneq=100, nStrongActIneq=20),
    dict(nVar=1, nObj=1, nIneq=3, nStrongActIneq=1),
    dict(nVar=50, nObj=50, nIneq=900, nStrongActIneq=10),  # C does not fit LDS: streamed variant
    # the n = 50 register-resident specialisation: no LDS extras (m <= 192), all 64 extras (m = 256), bounds + equalities
    dict(nVar=50, nObj=50, nIneq=150, nStrongActIneq=12),
    dict(nVar=50, nObj=50, nIneq=256, nStrongActIneq=20),
    dict(nVar=50, nObj=50, nEq=4, nIneq=196, nStrongActIneq=10, bounds=True, nStrongActBounds=5, doubleSidedIneq=True),
    # the n = 64 register-resident specialisation (two workgroups per CU): extras in LDS (m = 256), bounds + equalities
    dict(nVar=64, nObj=64, nIneq=128, nStrongActIneq=19),
    dict(nVar=64, nObj=64, nIneq=256, nStrongActIneq=25),
    dict(nVar=64, nObj=64, nEq=5, nIneq=190, nStrongActIneq=12, bounds=True, nStrongActBounds=6, doubleSidedIneq=True),
    # the register-resident instantiations WITHOUT general constraints (m = 0: C is an empty array / null pointer; the
    # reference's benchmark families test4 and test7, benchmarks/Solvers.cpp:625-639)
    dict(nVar=50, nObj=50),
    dict(nVar=50, nObj=50, bounds=True, nStrongActBounds=20),
    dict(nVar=64, nObj=64, bounds=True, nStrongActBounds=10),
    dict(nVar=50, nObj=50, nEq=20),
    # n = 60 (a size of the reference's benchmark families test1 / test3): its own register-resident instantiation
    dict(nVar=60, nObj=60, nIneq=120, nStrongActIneq=18),
    dict(nVar=60, nObj=60, nEq=12, nIneq=60, nStrongActIneq=18, bounds=True, nStrongActBounds=6, doubleSidedIneq=True),
    dict(nVar=60, nObj=60, nEq=24),
    # the other register-resident instantiations (n = 10, 16, 20, 30, 32, 40; up to five workgroups per CU)
    dict(nVar=16, nObj=16, nEq=2, nIneq=32, nStrongActIneq=4, bounds=True, nStrongActBounds=2, doubleSidedIneq=True),
    dict(nVar=20, nObj=20, nIneq=40, nStrongActIneq=6),
    dict(nVar=20, nObj=20, nEq=4, nIneq=250, nStrongActIneq=5, bounds=True, nStrongActBounds=3),  # extra columns in LDS
    dict(nVar=30, nObj=30, nEq=3, nIneq=60, nStrongActIneq=9, bounds=True, nStrongActBounds=3, doubleSidedIneq=True),
    dict(nVar=32, nObj=32, nIneq=64, nStrongActIneq=10),
    dict(nVar=40, nObj=40, nEq=8, nIneq=40, nStrongActIneq=12, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),
    dict(nVar=40, nObj=40, nIneq=300, nStrongActIneq=12),  # 108 LDS-staged columns: two passes of the second scan
    dict(nVar=30, nObj=30, nEq=3, nIneq=420, nStrongActIneq=9, bounds=True, nStrongActBounds=3, doubleSidedIneq=True),  # 228 staged columns, four passes
    dict(nVar=50, nObj=50, nEq=3, nIneq=420, nStrongActIneq=14, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),  # too many to stage: C streamed
    dict(nVar=10, nObj=10, bounds=True, nStrongActBounds=4),
    # sizes WITHOUT an instantiation: a cold solve runs embedded in the next instantiated size ([G 0; 0 I], zero rows in
    # C); the results must still be the oracle's, bit for bit
    dict(nVar=12, nObj=12, nEq=2, nIneq=24, nStrongActIneq=3, bounds=True, nStrongActBounds=2),
    dict(nVar=21, nObj=21, nIneq=42, nStrongActIneq=6, doubleSidedIneq=True),
    dict(nVar=45, nObj=45, nEq=5, nIneq=90, nStrongActIneq=12, bounds=True, nStrongActBounds=5, doubleSidedIneq=True),
    dict(nVar=55, nObj=55, nIneq=110, nStrongActIneq=16),
    dict(nVar=62, nObj=62, nEq=6, nIneq=200, nStrongActIneq=15, bounds=True, nStrongActBounds=6),
    dict(nVar=9, nObj=9, nIneq=18, nStrongActIneq=3),
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_reference_characteristic_sets(capi, P, oracle, ci):
    """The reference's "Random problems" characteristic sets (tests/GoldfarbIdnaniSolverTest.cpp:75-99) and wider
    ones: equalities (K12), bounds, double-sided constraints, n = 1, n = 64, C streamed from HBM."""
    ch = P.Characteristics(**CASES[ci])
    pb = P.generate_batch(48, ch, first_index=500 * ci)
    hb = ch.bounds
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None, nthreads=8)
    gpu = run_gpu(capi, pb, use_bounds=hb)
    assert_bit_identical(gpu, ref, f"case {ci}")
    assert np.all(ref["term"] == 0)


def test_termination_codes_and_edge_cases(capi, P, oracle):
    """NON_POS_HESSIAN, INFEASIBLE, MAX_ITER_REACHED, OVERCONSTRAINED and a no-constraint problem in one batch."""
    n, m = 4, 6
    B = 6
    rng = np.random.default_rng(5)
    G = np.tile(np.eye(n), (B, 1, 1))
    a = rng.standard_normal((B, n))
    C = rng.standard_normal((B, m, n))
    bl = np.full((B, m), -np.inf)
    bu = np.full((B, m), 10.0)
    G[0, 2, 2] = -1.0  # not positive definite
    C[1, 0] = [1, 0, 0, 0]
    C[1, 1] = [-1, 0, 0, 0]
    bu[1, 0] = -1.0
    bu[1, 1] = -1.0  # x0 <= -1 and -x0 <= -1: infeasible
    bu[2] = -5.0  # many violated constraints; with max_iter = 2 -> MAX_ITER_REACHED for all that need more
    bl[3] = bu[3] = 0.5  # six equalities for four unknowns: over-constrained
    G[4, 0, 0] = 0.0  # failure at column 0
    pb = P.PackedBatch(n=n, m=m, nb=0, G=G, a=a, C=C, bl=bl, bu=bu, xl=np.zeros((B, 0)), xu=np.zeros((B, 0)),
                       x_star=np.zeros((B, n)), lam_star=np.zeros((B, m)))
    ref = oracle.solve_batch(G, a, C, bl, bu, max_iter=2)
    gpu = run_gpu(capi, pb, max_iter=2)
    assert_bit_identical(gpu, ref, "edge cases")
    assert ref["term"][0] == 2 and ref["term"][1] == 3 and ref["term"][3] == 6 and ref["term"][4] == 2
    assert 4 in ref["term"]


def test_termination_codes_embedded_size(capi, P, oracle):
    """The same termination codes for a size that runs EMBEDDED in a larger instantiation (n = 12 inside n = 16): the
    embedding must not change what is reported (OVERCONSTRAINED counts the caller's unknowns, a failing pivot keeps its
    column index), and the factor handed back has the caller's shape."""
    n, m = 12, 14
    B = 6
    rng = np.random.default_rng(11)
    G = np.tile(np.eye(n), (B, 1, 1))
    a = rng.standard_normal((B, n))
    C = rng.standard_normal((B, m, n))
    bl = np.full((B, m), -np.inf)
    bu = np.full((B, m), 10.0)
    G[0, 7, 7] = -1.0  # not positive definite (column 7)
    C[1, 0] = 0.0
    C[1, 1] = 0.0
    C[1, 0, 3] = 1.0
    C[1, 1, 3] = -1.0
    bu[1, 0] = -1.0
    bu[1, 1] = -1.0  # x3 <= -1 and -x3 <= -1: infeasible
    bu[2] = -5.0  # many violated constraints; with max_iter = 2 -> MAX_ITER_REACHED
    bl[3, :13] = bu[3, :13] = 0.5  # thirteen equalities for twelve unknowns: over-constrained (but fewer than 16)
    G[4, 0, 0] = 0.0  # failure at column 0
    pb = P.PackedBatch(n=n, m=m, nb=0, G=G, a=a, C=C, bl=bl, bu=bu, xl=np.zeros((B, 0)), xu=np.zeros((B, 0)),
                       x_star=np.zeros((B, n)), lam_star=np.zeros((B, m)))
    ref = oracle.solve_batch(G, a, C, bl, bu, max_iter=2)
    gpu = run_gpu(capi, pb, max_iter=2, want_factor=True)
    assert_bit_identical(gpu, ref, "edge cases, embedded size")
    assert ref["term"][0] == 2 and ref["term"][1] == 3 and ref["term"][3] == 6 and ref["term"][4] == 2
    assert 4 in ref["term"]
    ok = ref["term"] != 2
    assert np.array_equal(np.tril(gpu["L"].transpose(0, 2, 1))[ok], np.tril(ref["L"].transpose(0, 2, 1))[ok])


def test_c3_full_bench_batch(capi, P, oracle, pkg):
    """The full BASELINE workload (B = 4096, n = 50, m = 200): bit-exact against the oracle on every problem, and the
    size-independent properties -- every solve reaches the planted optimum, the active set is the planted one, the
    multipliers have the right signs."""
    import os

    pb = P.generate_batch(4096, P.benchmark_characteristics(50, 200))
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=min(64, len(os.sched_getaffinity(0))))
    gpu = run_gpu(capi, pb)
    assert_bit_identical(gpu, ref, "c3 x 4096")
    assert np.all(gpu["term"] == 0)
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6
    assert np.array_equal(gpu["status"] == 2, pb.lam_star > 0)  # UPPER exactly where the planted multiplier is positive
    assert np.all(gpu["u"] >= 0) and np.max(np.abs(gpu["u"] - pb.lam_star)) <= 1e-6
    assert 25 < gpu["iters"].mean() < 40
    # the same batch again, several launches back to back on a full chip (three workgroups per CU, every LDS slot re-used by a
    # later workgroup): the hand-off words of the role-specialised loop must not depend on what a predecessor left behind
    import torch

    dev = capi.to_device(pb, "cuda:0")
    s = capi.BatchSolver(pb.n, pb.m, False, pb.B, device="cuda:0")
    for _ in range(6):
        r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
    torch.cuda.synchronize()
    assert np.array_equal(r.x.cpu().numpy(), ref["x"]) and np.array_equal(r.iters.cpu().numpy(), ref["iters"])
    assert np.array_equal(r.u.cpu().numpy(), ref["u"]) and np.array_equal(r.status.cpu().numpy(), ref["status"])


def test_trace_matches_oracle(capi, P, oracle):
    pb = P.generate_batch(8, P.benchmark_characteristics(50, 200), first_index=31)
    gpu = run_gpu(capi, pb, trace_cap=128)
    for b in range(8):
        r = oracle.solve(pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], trace_cap=128)
        tr = gpu["trace"][b][: r.iters]
        got = [(int(t["p"]), int(t["status"]), int(t["l"]), int(t["kind"]), float(t["t1"]), float(t["t2"])) for t in tr]
        assert got == r.trace


def test_randomised_shape_sweep(capi, P, oracle):
    """Fuzz over shapes and constraint mixes (n = 1 .. 64, equalities, bounds, double-sided, weakly active =
    degenerate problems on which the reference algorithm may legitimately stop with INFEASIBLE): whatever the
    oracle does, the HIP path does bit for bit."""
    rng = np.random.default_rng(20260301)
    seen_terms = set()
    for trial in range(40):
        n = int(rng.choice([1, 2, 3, 7, 16, 17, 31, 32, 33, 48, 50, 63, 64]))
        nEq = int(rng.integers(0, max(1, n // 4) + 1)) if rng.random() < 0.4 else 0
        bounds = bool(rng.random() < 0.4)
        nIneq = int(rng.integers(0, 4 * n + 8))
        room = n - nEq
        nSA = int(rng.integers(0, min(nIneq, room) + 1))
        nSB = int(rng.integers(0, max(0, min(n, room - nSA)) + 1)) if bounds else 0
        nWA = int(rng.integers(0, min(3, nIneq - nSA) + 1)) if rng.random() < 0.3 else 0
        nWB = int(rng.integers(0, min(2, n - nSB) + 1)) if (bounds and rng.random() < 0.3) else 0
        ch = P.Characteristics(nVar=n, nObj=n, nEq=nEq, nIneq=nIneq, nStrongActIneq=nSA, nWeakActIneq=nWA, bounds=bounds,
                               nStrongActBounds=nSB, nWeakActBounds=nWB, doubleSidedIneq=bool(rng.random() < 0.5))
        pb = P.generate_batch(24, ch, first_index=100000 + 50 * trial)
        ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if bounds else None, pb.xu if bounds else None, nthreads=8)
        gpu = run_gpu(capi, pb, use_bounds=bounds)
        assert_bit_identical(gpu, ref, f"trial {trial}: {ch}")
        seen_terms |= set(ref["term"].tolist())
    assert 0 in seen_terms


def test_box_and_single_constraint_solver(pkg, oracle):
    """jrlqp_box_solve_batch (closed-form init_ of src/experimental/BoxAndSingleConstraintSolver.cpp:46-94 as the kernel
    prologue) == the oracle's restatement of that solver, bit for bit (status, iterations, active set, x, u, f), on the
    reference's own test set-up (tests/BoxAndSingleConstraintSolverTest.cpp:16-47: n = 10, 11 problems with the constraint
    inactive + 11 active, generator of BoxAndSingleConstraintSolver.cpp:99-176) and on other sizes; and, as in that test,
    == the plain GoldfarbIdnaniSolver on (G = I, a = -x0) up to the constant 1/2 ||x0||^2."""
    import torch

    import jrl_qp_amd.capi as capi
    import jrl_qp_amd.problems as P

    dev = lambda v: torch.tensor(np.ascontiguousarray(v), device="cuda")
    # (beyond 64 variables -- round 4 -- the closed form runs in the large-n kernel, thread = variable: R in LDS up to 104, in the
    # workspace beyond, J in the workspace past 128)
    for n, reps, seed in ((10, 11, 20201), (1, 4, 5), (7, 6, 6), (33, 6, 7), (64, 5, 8), (65, 4, 9), (100, 5, 10), (120, 3, 11), (200, 3, 12), (256, 2, 13)):
        rng = np.random.default_rng(seed)
        pbs = [P.box_and_single_constraint_problem(n, act, rng) for act in (False, True) for _ in range(reps)]
        B = len(pbs)
        x0, c, bl, xl, xu = (np.stack([np.atleast_1d(p[k]) for p in pbs]) for k in range(5))
        s = capi.BoxAndSingleConstraintSolver(n, B)
        r = s.solve(dev(x0), dev(c), dev(bl), dev(xl), dev(xu))
        torch.cuda.synchronize()
        x, u, f = r.x.cpu().numpy(), r.u.cpu().numpy(), r.f.cpu().numpy()
        term, iters, status = r.term.cpu().numpy(), r.iters.cpu().numpy(), r.status.cpu().numpy()
        nact = 0
        for b in range(B):
            ref = oracle.box_solve(x0[b], c[b], bl[b, 0], xl[b], xu[b])
            what = f"n={n} problem {b}"
            assert term[b] == ref.term == 0, what
            assert iters[b] == ref.iters, what
            assert np.array_equal(status[b], ref.status), what
            assert np.array_equal(x[b], ref.x) and np.array_equal(u[b], ref.u) and f[b] == ref.f, what
            gi = oracle.solve(np.eye(n), -x0[b], c[b].reshape(n, 1), bl[b], [np.inf], xl[b], xu[b])
            assert np.allclose(x[b], gi.x, rtol=1e-12, atol=1e-14)
            assert abs(f[b] - (gi.f + 0.5 * x0[b] @ x0[b])) <= 1e-9 * (1 + abs(f[b]))
            nact += int(status[b, 0] != 0)
        assert nact <= reps and (n < 7 or nact == reps)  # the "active" half ends with the general constraint active
    # beyond the batched kernels' 256 unknowns: refused, never a silent wrong answer; and the plain entry point past n = 104 asks for
    # the workspace instead of allocating one
    with pytest.raises(capi.JrlqpError) as ei:
        capi.BoxAndSingleConstraintSolver(257, 2)
    assert ei.value.status == capi.ERR_UNSUPPORTED_SIZE
    lib = capi.load_library()
    z = torch.zeros((2, 120), dtype=torch.float64, device="cuda")
    o3 = [torch.zeros((2, 121), dtype=torch.float64, device="cuda"), torch.zeros((2, 121), dtype=torch.int32, device="cuda")]
    i1 = [torch.zeros(2, dtype=torch.int32, device="cuda") for _ in range(2)]
    f1 = torch.zeros(2, dtype=torch.float64, device="cuda")
    P_ = lambda t: ct.c_void_p(t.data_ptr())
    rc = lib.jrlqp_box_solve_batch(120, 2, P_(z), P_(z), P_(f1), P_(z), P_(z), None, P_(z.clone()), P_(o3[0]), P_(o3[1]), P_(i1[0]), P_(i1[1]), P_(f1.clone()),
                                   None, 0, None)
    assert rc == capi.ERR_WORKSPACE
    # termination statuses other than SUCCESS come out of the same loop: infeasible constraint, iteration limit
    n = 5
    x0 = np.array([[0.5, -2.0, 3.0, 0.1, -0.7]] * 2)
    xl = np.array([[0.0, -1.0, -1.0, -1.0, -0.5]] * 2)
    xu = np.array([[1.0, 1.0, 1.0, 1.0, 0.5]] * 2)
    c = np.ones((2, n))
    bl = np.array([100.0, 3.9])
    for max_iter in (500, 0):
        s = capi.BoxAndSingleConstraintSolver(n, 2, max_iter=max_iter)
        r = s.solve(dev(x0), dev(c), dev(bl), dev(xl), dev(xu))
        torch.cuda.synchronize()
        for b in range(2):
            ref = oracle.box_solve(x0[b], c[b], bl[b], xl[b], xu[b], max_iter=max_iter)
            assert int(r.term[b]) == ref.term and int(r.iters[b]) == ref.iters
            assert np.array_equal(r.x[b].cpu().numpy(), ref.x) and np.array_equal(r.u[b].cpu().numpy(), ref.u)
            assert float(r.f[b]) == ref.f and np.array_equal(r.status[b].cpu().numpy(), ref.status)
        if max_iter == 500:
            assert int(r.term[0]) == 3 and int(r.term[1]) == 0
        else:
            assert int(r.term[0]) == 4


@pytest.mark.parametrize("n,m", [(50, 200), (40, 80), (64, 300), (45, 120)])
def test_iteration_limit_in_the_role_specialised_loop(capi, P, oracle, n, m):
    """MAX_ITER_REACHED in the hand-off loop of the n >= 40 kernels (every way out of that loop has to release the helper
    waves): limits 0, 1, 2, 7 and one beyond what the problems need, with the per-iteration trace."""
    pb = P.generate_batch(48, P.benchmark_characteristics(n, m), first_index=31000 + n)
    for max_iter in (0, 1, 2, 7, 500):
        ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, max_iter=max_iter)
        gpu = run_gpu(capi, pb, max_iter=max_iter, trace_cap=8)
        assert_bit_identical(gpu, ref, f"n={n} max_iter={max_iter}")
        if max_iter < 8:
            assert np.all(gpu["term"] == 4) or max_iter >= int(ref["iters"].max())
    r0 = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=7, trace_cap=8)
    g = run_gpu(capi, pb, max_iter=7, trace_cap=8)
    for k, rec in enumerate(r0.trace):
        t = g["trace"][0, k]
        assert (int(t["p"]), int(t["status"]), int(t["l"]), int(t["kind"]), float(t["t1"]), float(t["t2"])) == rec
