"""GPU parity of the batched path for 64 < n <= 128 (gi_batch_big.hip through jrlqp_gi_solve_batch): the sizes the
reference's own benchmark families reach (benchmarks/Solvers.cpp:613-623, n = 70 ... 100) and the ABI's ceiling n = 128
-- termination status, iteration count, activation statuses, x, u, f and the Cholesky factor, bit for bit against the CPU
oracle; equalities, bounds, double-sided constraints, the other termination statuses and more problems than workgroups
(the grid is persistent) included."""
import numpy as np
import pytest

from test_gpu_batch import assert_bit_identical, run_gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


BIG_CASES = [
    dict(nVar=70, nObj=70, nIneq=140, nStrongActIneq=21),  # reference families test1/test3 sizes (30 % active)
    dict(nVar=80, nObj=80, nIneq=160, nStrongActIneq=40),
    dict(nVar=100, nObj=100, nIneq=200, nStrongActIneq=30),
    dict(nVar=128, nObj=128, nIneq=256, nStrongActIneq=38),  # ABI ceiling: R no longer fits in LDS (scratch workspace)
    dict(nVar=65, nObj=65, nEq=6, nIneq=90, nStrongActIneq=12),  # equalities (initActiveSet)
    dict(nVar=90, nObj=90, nEq=4, nIneq=100, nStrongActIneq=20, bounds=True, nStrongActBounds=9),  # + bounds
    dict(nVar=110, nObj=110, nIneq=130, nStrongActIneq=25, doubleSidedIneq=True, nWeakActIneq=6),  # double-sided, weakly active
    dict(nVar=72, nObj=72),  # unconstrained
    dict(nVar=128, nObj=128, nEq=10, nIneq=60, nStrongActIneq=15, bounds=True, nStrongActBounds=20, nWeakActBounds=4),
]


@pytest.mark.parametrize("ci", range(len(BIG_CASES)))
def test_big_n_matches_oracle(capi, P, oracle, ci):
    ch = P.Characteristics(**BIG_CASES[ci])
    pb = P.generate_batch(6, ch, first_index=7000 + 20 * ci)
    hb = ch.bounds
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None, nthreads=6)
    gpu = run_gpu(capi, pb, use_bounds=hb, want_factor=True)
    assert_bit_identical(gpu, ref, f"big case {ci}")
    assert np.all(ref["term"] == 0)
    assert np.array_equal(np.tril(gpu["L"].transpose(0, 2, 1)), np.tril(ref["L"].transpose(0, 2, 1)))
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6 * max(1.0, np.max(np.linalg.norm(pb.x_star, axis=1)))
    assert int(ref["iters"].max()) > 0 or ch.nIneq + ch.nEq == 0


WIDE_CASES = [
    dict(nVar=129, nObj=129, nIneq=200, nStrongActIneq=40),  # the first size whose J lives in the workspace
    dict(nVar=160, nObj=160, nEq=8, nIneq=150, nStrongActIneq=30, bounds=True, nStrongActBounds=12),
    dict(nVar=200, nObj=200, nIneq=400, nStrongActIneq=60, doubleSidedIneq=True, nWeakActIneq=5),
    dict(nVar=256, nObj=256, nIneq=300, nStrongActIneq=77),  # the batched ceiling
    dict(nVar=256, nObj=256, nEq=20, nIneq=120, nStrongActIneq=30, bounds=True, nStrongActBounds=40, nWeakActBounds=3),
    dict(nVar=192, nObj=192),  # unconstrained
]


@pytest.mark.parametrize("ci", range(len(WIDE_CASES)))
def test_wide_n_matches_oracle(capi, P, oracle, ci):
    """128 < n <= 256 (gi_batch_big.hip, WIDE: L / J and R in the caller's workspace, four vector entries per lane): status,
    iterations, active set, x, u, f and the Cholesky factor bit for bit against the oracle."""
    ch = P.Characteristics(**WIDE_CASES[ci])
    pb = P.generate_batch(5, ch, first_index=8000 + 20 * ci)
    hb = ch.bounds
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if hb else None, pb.xu if hb else None, nthreads=5)
    gpu = run_gpu(capi, pb, use_bounds=hb, want_factor=True)
    assert_bit_identical(gpu, ref, f"wide case {ci}")
    assert np.all(ref["term"] == 0)
    assert np.array_equal(np.tril(gpu["L"].transpose(0, 2, 1)), np.tril(ref["L"].transpose(0, 2, 1)))
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6 * max(1.0, np.max(np.linalg.norm(pb.x_star, axis=1)))


def test_wide_n_grid_and_limits(capi, P, oracle):
    """More problems than resident workgroups at n = 140 (persistent grid: 3 per CU), the iteration limit, and the size beyond
    the ceiling: n = 257 is refused."""
    import torch

    pb = P.generate_batch(1100, P.benchmark_characteristics(140, 160), first_index=9300)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
    assert_bit_identical(run_gpu(capi, pb), ref, "n=140 x 1100")
    small = P.generate_batch(4, P.benchmark_characteristics(150, 200), first_index=9400)
    for max_iter in (0, 3):
        ref = oracle.solve_batch(small.G, small.a, small.C, small.bl, small.bu, max_iter=max_iter)
        assert_bit_identical(run_gpu(capi, small, max_iter=max_iter), ref, f"n=150 max_iter={max_iter}")
    with pytest.raises(capi.JrlqpError):
        capi.BatchSolver(257, 10, False, 2, device="cuda:0")
    torch.cuda.synchronize()


def test_big_n_between_79_and_95_with_and_without_a_workspace(capi, P, oracle):
    """n = 79 .. 95: J and the packed R fit once in a CU's LDS, J alone twice.  A batch of more problems than the device has CUs, with
    the caller's workspace (what jrlqp_gi_batch_workspace_bytes asks for since round 5), keeps R there and two workgroups share a CU;
    a smaller batch, or a caller that passes no workspace -- the contract of the entries without `_ws`: none is needed while R fits
    in LDS -- runs with R in LDS, one workgroup per CU.  Same bits every way."""
    import torch

    ch = P.Characteristics(nVar=84, nObj=84, nEq=3, nIneq=130, nStrongActIneq=22, bounds=True, nStrongActBounds=6, doubleSidedIneq=True)
    base = P.generate_batch(5, ch, first_index=9300)
    ref5 = oracle.solve_batch(base.G, base.a, base.C, base.bl, base.bu, base.xl, base.xu, nthreads=5)
    assert np.all(ref5["term"] == 0) and int(ref5["iters"].max()) > 10
    assert_bit_identical(run_gpu(capi, base, use_bounds=True), ref5, "n = 84, five problems")
    B = 5 * 120  # more problems than CUs: two workgroups per CU
    tile = lambda a: np.ascontiguousarray(np.concatenate([a] * 120))
    pb = P.PackedBatch(n=base.n, m=base.m, nb=base.nb, G=tile(base.G), a=tile(base.a), C=tile(base.C), bl=tile(base.bl), bu=tile(base.bu),
                       xl=tile(base.xl), xu=tile(base.xu), x_star=tile(base.x_star), lam_star=tile(base.lam_star))
    ref = {k: np.concatenate([v] * 120) for k, v in ref5.items() if k != "L"}
    assert_bit_identical(run_gpu(capi, pb, use_bounds=True), ref, "n = 84 x 600 with the workspace")
    dev = capi.to_device(pb, "cuda:0")
    s = capi.BatchSolver(pb.n, pb.m, True, B, device="cuda:0")
    assert s.workspace_bytes > 0
    s.workspace, s.workspace_bytes = None, 0
    r = s.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], dev["xl"], dev["xu"])
    torch.cuda.synchronize()
    out = dict(x=r.x.cpu().numpy(), u=r.u.cpu().numpy(), status=r.status.cpu().numpy(), term=r.term.cpu().numpy(), iters=r.iters.cpu().numpy(),
               f=r.f.cpu().numpy())
    assert_bit_identical(out, ref, "n = 84 x 600 without a workspace")


def test_big_n_persistent_grid_more_problems_than_cus(capi, P, oracle):
    """600 problems at n = 70 on a 256-CU grid: every workgroup solves two or three problems back to back."""
    pb = P.generate_batch(600, P.benchmark_characteristics(70, 100), first_index=9100)
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=8)
    gpu = run_gpu(capi, pb)
    assert_bit_identical(gpu, ref, "n=70 x 600")
    assert np.all(gpu["term"] == 0)


def test_big_n_status_codes_and_trace(capi, P, oracle):
    n, m = 66, 6
    rng = np.random.default_rng(11)
    a = rng.standard_normal(n)
    C = rng.standard_normal((m, n))
    bl = np.full(m, -np.inf)
    mk = lambda G_, bu_, C_=C: P.PackedBatch(n=n, m=m, nb=0, G=G_[None], a=a[None], C=C_[None], bl=bl[None], bu=bu_[None], xl=np.zeros((1, 0)),
                                             xu=np.zeros((1, 0)), x_star=np.zeros((1, n)), lam_star=np.zeros((1, m)))
    G = np.eye(n)
    # MAX_ITER_REACHED, with the per-iteration trace
    pb = mk(G, np.full(m, -5.0))
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, max_iter=2)
    gpu = run_gpu(capi, pb, max_iter=2, trace_cap=4)
    assert_bit_identical(gpu, ref, "max iter")
    assert gpu["term"][0] == 4 and gpu["iters"][0] == 2
    r1 = oracle.solve(G, a, C.T, bl, np.full(m, -5.0), max_iter=2, trace_cap=4)
    for k, rec in enumerate(r1.trace):
        t = gpu["trace"][0, k]
        assert (int(t["p"]), int(t["status"]), int(t["l"]), int(t["kind"]), float(t["t1"]), float(t["t2"])) == rec
    # NON_POS_HESSIAN (first and a later column)
    for col in (0, 40):
        Gb = np.eye(n)
        Gb[col, col] = -1.0
        pbn = mk(Gb, np.full(m, -5.0))
        refn = oracle.solve_batch(pbn.G, pbn.a, pbn.C, pbn.bl, pbn.bu)
        gpun = run_gpu(capi, pbn)
        assert gpun["term"][0] == refn["term"][0] == 2
    # INFEASIBLE: x_0 >= 1 and x_0 <= -1
    C2 = C.copy()
    C2[0] = 0.0
    C2[0, 0] = 1.0
    C2[1] = 0.0
    C2[1, 0] = -1.0
    bu2 = np.full(m, 10.0)
    bu2[0] = bu2[1] = -1.0
    pbi = mk(G, bu2, C2)
    refi = oracle.solve_batch(pbi.G, pbi.a, pbi.C, pbi.bl, pbi.bu)
    gpui = run_gpu(capi, pbi)
    assert_bit_identical(gpui, refi, "infeasible")
    assert refi["term"][0] == 3


BIG_WARM = [
    dict(nVar=70, nObj=70, nIneq=140, nStrongActIneq=20),
    dict(nVar=70, nObj=70, nEq=6, nIneq=60, nStrongActIneq=14, bounds=True, nStrongActBounds=6, doubleSidedIneq=True),
    dict(nVar=88, nObj=88, nEq=5, nIneq=120, nStrongActIneq=25, bounds=True, nStrongActBounds=7),  # 79 .. 95: two workgroups per CU, the packed R in the workspace
    dict(nVar=100, nObj=100, nIneq=200, nStrongActIneq=30),
    dict(nVar=100, nObj=100, nEq=10, nIneq=90, nStrongActIneq=20, bounds=True, nStrongActBounds=8),
    dict(nVar=128, nObj=128, nIneq=160, nStrongActIneq=36),  # R and B = L^-1 N both in the caller's workspace
    dict(nVar=128, nObj=128, nEq=12, nIneq=100, nStrongActIneq=20, bounds=True, nStrongActBounds=10, doubleSidedIneq=True),
    dict(nVar=130, nObj=130, nIneq=150, nStrongActIneq=40),  # beyond 128: L / J, R and B all in the workspace, four entries per lane
    dict(nVar=200, nObj=200, nEq=10, nIneq=150, nStrongActIneq=40, bounds=True, nStrongActBounds=20, doubleSidedIneq=True),
    dict(nVar=256, nObj=256, nIneq=280, nStrongActIneq=70),
]


@pytest.mark.parametrize("ci", range(len(BIG_WARM)))
def test_big_n_warm_start_bit_identical(capi, P, oracle, ci):
    """jrlqp_gi_solve_batch_warm for 64 < n <= 128 (experimental::GoldfarbIdnaniSolver, src/experimental/GoldfarbIdnaniSolver.cpp:
    20-112, 306-487; the sizes of the reference's warm-start benchmark, benchmarks/SolversWarmStart.cpp:379-380, go to n = 100):
    no guess, the exact active set (0 iterations: tests/GoldfarbIdnaniSolverTest.cpp:176,181), stale guesses (purge loop) and an
    all-lower guess (more than n guesses: trimmed from the back) -- every output bit-identical to the oracle's orc_gi_solve_warm."""
    from test_gpu_warm import oracle_cold, oracle_warm, perturbed_guess, run_warm

    ch = P.Characteristics(**BIG_WARM[ci])
    pb = P.generate_batch(12 if ch.nVar <= 128 else 5, ch, first_index=5200 + 40 * ci, seed0=0xB16)
    hb = ch.bounds
    cold = oracle_cold(oracle, pb, hb)
    rng = np.random.default_rng(100 + ci)
    L, LB = 1, 4
    guesses = {
        "none": None,
        "exact": cold["status"],
        "stale": perturbed_guess(rng, cold["status"], pb.m),
        "very stale": perturbed_guess(rng, cold["status"], pb.m, flip=0.5, add=0.2),
        "everything lower": np.where(np.arange(pb.m + pb.nb)[None, :] < pb.m, L, LB).repeat(pb.B, 0).astype(np.int32),
    }
    purges = 0
    for name, g in guesses.items():
        ref = oracle_warm(oracle, pb, g, hb)
        gpu = run_warm(capi, pb, g, hb)
        assert_bit_identical(gpu, ref, f"{BIG_WARM[ci]}, guess '{name}'")
        if name == "exact" and ch.nEq == 0:
            assert np.all(ref["iters"][ref["term"] == 0] == 0)  # the reference's warm-start expectation
        purges += int(np.count_nonzero(ref["iters"]))
    assert purges > 0
    # "previous": the solver's own last active set, in place (the reference's as.empty() case)
    gpu = run_warm(capi, pb, "previous", hb)
    ref = oracle_warm(oracle, pb, cold["status"], hb)
    assert_bit_identical(gpu, ref, "previous")


def test_big_n_warm_error_paths(capi, P, oracle):
    """OVERCONSTRAINED (reported before the factorisation), NON_POS_HESSIAN and the iteration limit inside / behind the purge loop."""
    from test_gpu_warm import oracle_cold, oracle_warm, perturbed_guess, run_warm

    ch = P.Characteristics(nVar=72, nObj=72, nEq=4, nIneq=90, nStrongActIneq=12, bounds=True, nStrongActBounds=4)
    pb = P.generate_batch(8, ch, first_index=77, seed0=0xE45)
    pb.bu[1, :73] = pb.bl[1, :73]  # 73 equalities > n
    pb.G[2, 0, 0] = -1.0
    pb.bu[3, :73] = pb.bl[3, :73]
    pb.G[3, 0, 0] = -1.0
    cold = oracle_cold(oracle, pb, True)
    g = perturbed_guess(np.random.default_rng(5), cold["status"], pb.m, flip=0.5, add=0.3)
    for max_iter in (500, 3, 0):
        ref = oracle_warm(oracle, pb, g, True, max_iter=max_iter)
        gpu = run_warm(capi, pb, g, True, max_iter=max_iter)
        assert_bit_identical(gpu, ref, f"max_iter={max_iter}")
    ref = oracle_warm(oracle, pb, g, True)
    assert ref["term"][1] == 6 and ref["term"][2] == 2 and ref["term"][3] == 6
