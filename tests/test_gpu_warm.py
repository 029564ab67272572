"""GPU parity of the warm-started batched path (jrlqp_gi_solve_batch_warm, through the C ABI) against the oracle's
orc_gi_solve_warm on the same seeded inputs and guesses: BIT-exact on termination status, iteration count, active
set, x, u and f (tolerance 0 ulp -- same CANON contract as the cold path), plus the reference's own "Warm-start"
expectations (tests/GoldfarbIdnaniSolverTest.cpp:127-210): a warm start from the optimal active set takes 0
iterations."""
import numpy as np
import pytest

from test_gpu_batch import assert_bit_identical

pytestmark = pytest.mark.gpu

I, L, U, E, LB, UB, F = range(7)


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


def run_warm(capi, pb, guess, use_bounds=False, max_iter=500, previous_of=None):
    """guess: (B, m+nb) int32 numpy array, None, or "previous" (then a cold solve runs first on the same solver)."""
    import torch

    dev = capi.to_device(pb, "cuda:0")
    s = capi.BatchSolver(pb.n, pb.m, use_bounds, pb.B, device="cuda:0", max_iter=max_iter)
    args = (dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"], dev["xl"] if use_bounds else None, dev["xu"] if use_bounds else None)
    if isinstance(guess, str):
        s.solve(*args)
        g = "previous"
    elif guess is None:
        g = None
    else:
        g = torch.from_numpy(np.ascontiguousarray(guess, dtype=np.int32)).to("cuda:0")
    r = s.solve_warm(*args, guess=g)
    torch.cuda.synchronize()
    return dict(x=r.x.cpu().numpy(), u=r.u.cpu().numpy(), status=r.status.cpu().numpy(), term=r.term.cpu().numpy(),
                iters=r.iters.cpu().numpy(), f=r.f.cpu().numpy())


def oracle_warm(oracle, pb, guess, use_bounds, max_iter=500):
    return oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if use_bounds else None, pb.xu if use_bounds else None,
                              max_iter=max_iter, nthreads=8, warm=True, guess=guess)


def oracle_cold(oracle, pb, use_bounds):
    return oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, pb.xl if use_bounds else None, pb.xu if use_bounds else None, nthreads=8)


REF_FAMILIES = [
    dict(nVar=5, nObj=5),
    dict(nVar=5, nObj=5, nEq=2),
    dict(nVar=5, nObj=5, nIneq=8, nStrongActIneq=4),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=3),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=1, bounds=True, nStrongActBounds=2),
]


@pytest.mark.parametrize("fi", range(len(REF_FAMILIES)))
def test_reference_warm_start_families(capi, P, oracle, fi):
    """The five families of the reference's Warm-start test, 512 draws each: warm start from the cold solution's
    active set -> 0 iterations, bit-identical to the oracle; then from the solver's own previous active set."""
    ch = P.Characteristics(**REF_FAMILIES[fi])
    pb = P.generate_batch(512, ch, first_index=7000 * fi, seed0=0x57A27)
    hb = ch.bounds
    if pb.m + pb.nb == 0:
        pytest.skip("no constraint: nothing to warm start")
    cold = oracle_cold(oracle, pb, hb)
    assert np.all(cold["term"] == 0)
    ref = oracle_warm(oracle, pb, cold["status"], hb)
    gpu = run_warm(capi, pb, cold["status"], hb)
    assert_bit_identical(gpu, ref, f"family {fi}")
    assert np.all(gpu["term"] == 0)
    assert np.count_nonzero(gpu["iters"]) <= max(1, pb.B // 1000 + 1)  # the reference tolerates < 0.1 % near-degenerate draws
    assert np.allclose(gpu["x"], pb.x_star, rtol=1e-6, atol=1e-7)
    again = run_warm(capi, pb, "previous", hb)  # cold solve, then warm solve reading the solver's own status buffer in place
    ref2 = oracle_warm(oracle, pb, cold["status"], hb)
    assert_bit_identical(again, ref2, f"family {fi} (previous active set)")


def test_warm_bench_shape_zero_iterations(capi, P, oracle, pkg):
    """n = 50, m = 200 (the register-resident instantiation): optimal active set as guess."""
    import jrl_qp_amd.kkt as kkt

    pb = P.generate_batch(256, P.benchmark_characteristics(50, 200))
    cold = oracle_cold(oracle, pb, False)
    ref = oracle_warm(oracle, pb, cold["status"], False)
    gpu = run_warm(capi, pb, cold["status"])
    assert_bit_identical(gpu, ref, "c3 warm")
    assert np.all(gpu["term"] == 0) and np.all(gpu["iters"] == 0)
    assert np.array_equal(gpu["status"], cold["status"])
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6
    assert np.allclose(gpu["u"], cold["u"], rtol=1e-6, atol=1e-8)
    for b in range(0, 256, 32):
        assert kkt.test_kkt(gpu["x"][b], gpu["u"][b], pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b])


WIDE = [
    dict(nVar=10, nObj=10, nEq=3, nIneq=20, nStrongActIneq=3, bounds=True, nStrongActBounds=2, doubleSidedIneq=True),
    dict(nVar=33, nObj=33, nEq=5, nIneq=70, nStrongActIneq=10, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),
    dict(nVar=64, nObj=64, nIneq=100, nStrongActIneq=20),
    dict(nVar=60, nObj=60, nEq=6, nIneq=90, nStrongActIneq=15, bounds=True, nStrongActBounds=5),  # the n = 60 instantiation, warm
    dict(nVar=20, nObj=20, nEq=2, nIneq=40, nStrongActIneq=6, bounds=True, nStrongActBounds=2),
    dict(nVar=45, nObj=45, nEq=5, nIneq=90, nStrongActIneq=12, bounds=True, nStrongActBounds=5, doubleSidedIneq=True),  # embedded in n = 50
    dict(nVar=12, nObj=12, nIneq=24, nStrongActIneq=4),  # embedded in n = 16
    dict(nVar=40, nObj=40, nIneq=80, nStrongActIneq=12, doubleSidedIneq=True),
    dict(nVar=1, nObj=1, nIneq=3, nStrongActIneq=1),
    dict(nVar=50, nObj=50, nIneq=900, nStrongActIneq=10),  # C streamed from L2/HBM
    dict(nVar=50, nObj=50, nIneq=256, nStrongActIneq=20),  # 64 LDS-staged extra columns
    dict(nVar=50, nObj=50, nEq=4, nIneq=196, nStrongActIneq=10, bounds=True, nStrongActBounds=5, doubleSidedIneq=True),
]


def perturbed_guess(rng, status, m, flip=0.15, add=0.05):
    """A plausible stale active set: some active inequalities forgotten or flipped to the other side, some inactive
    ones wrongly activated (general constraints get LOWER/UPPER, bounds LOWER_BOUND/UPPER_BOUND)."""
    g = status.copy()
    B, mt = g.shape
    r = rng.random(g.shape)
    is_bound = np.arange(mt)[None, :] >= m
    act = (g != I) & (g != E) & (g != F)
    g[act & (r < flip / 2)] = I
    sw = act & (r >= flip / 2) & (r < flip)
    g[sw & (status == L)] = U
    g[sw & (status == U)] = L
    g[sw & (status == LB)] = UB
    g[sw & (status == UB)] = LB
    new = (status == I) & (r < add)
    side = rng.integers(0, 2, g.shape)
    g[new & ~is_bound] = (L + side)[new & ~is_bound]
    g[new & is_bound] = (LB + side)[new & is_bound]
    return g.astype(np.int32)


@pytest.mark.parametrize("ci", range(len(WIDE)))
def test_stale_guesses_bit_identical(capi, P, oracle, ci):
    """Wrong guesses exercise the purge loop (u < -1e-14 -> drop + re-initialisation) and the hand-over to the
    regular loop; none, exact, stale and all-active guesses."""
    ch = P.Characteristics(**WIDE[ci])
    pb = P.generate_batch(48, ch, first_index=900 * ci, seed0=0xFACE)
    hb = ch.bounds
    cold = oracle_cold(oracle, pb, hb)
    rng = np.random.default_rng(ci)
    guesses = {
        "none": None,
        "exact": cold["status"],
        "stale": perturbed_guess(rng, cold["status"], pb.m),
        "very stale": perturbed_guess(rng, cold["status"], pb.m, flip=0.6, add=0.3),
        "everything lower": np.where(np.arange(pb.m + pb.nb)[None, :] < pb.m, L, LB).repeat(pb.B, 0).astype(np.int32),
    }
    purges = 0
    for name, g in guesses.items():
        ref = oracle_warm(oracle, pb, g, hb)
        gpu = run_warm(capi, pb, g, hb)
        assert_bit_identical(gpu, ref, f"case {ci}, guess '{name}'")
        ok = ref["term"] == 0
        assert ok.mean() > 0.8, (name, np.bincount(ref["term"]))
        # Same optimum as the cold solve -- except that a wrong guess can go wrong when equalities are present: the
        # step-length test looks the status up by POSITION (src/experimental/GoldfarbIdnaniSolver.cpp:191, same as
        # GoldfarbIdnaniSolver.cpp:162) and the warm start orders bounds before equalities, so a guessed bound sitting
        # at the position of an equality's index is never dropped (the reference's test notes "this fails in some
        # circumstances", tests/GoldfarbIdnaniSolverTest.cpp:186).  Reproduced bit for bit, not "fixed".
        if ch.nEq == 0 or name in ("none", "exact"):
            assert np.allclose(gpu["x"][ok], cold["x"][ok], rtol=1e-6, atol=1e-7)
        purges += int(np.count_nonzero(ref["iters"]))
    assert purges > 0


def test_warm_error_paths_and_limits(capi, P, oracle):
    """OVERCONSTRAINED (more than n equalities, reported before the factorisation), NON_POS_HESSIAN, max_iter reached
    inside / right after the purge loop, ill-typed guess entries."""
    ch = P.Characteristics(nVar=6, nObj=6, nEq=2, nIneq=12, nStrongActIneq=2, bounds=True, nStrongActBounds=1)
    pb = P.generate_batch(16, ch, first_index=42, seed0=0xE44)
    # problem 1: 7 equalities; problem 2: indefinite G; problem 3: both (OVERCONSTRAINED wins)
    pb.bu[1, :7] = pb.bl[1, :7]
    pb.G[2, 0, 0] = -1.0
    pb.bu[3, :7] = pb.bl[3, :7]
    pb.G[3, 0, 0] = -1.0
    rng = np.random.default_rng(3)
    cold = oracle_cold(oracle, pb, True)
    g = perturbed_guess(rng, cold["status"], pb.m, flip=0.5, add=0.4)
    g[4, :] = 9  # out of the enum
    g[5, : pb.m] = UB  # bound statuses on general constraints
    g[5, pb.m:] = U  # and general ones on bounds
    g[6, :] = E  # EQUALITY guesses on general constraints are taken, on bounds ignored
    for max_iter in (500, 2, 0):
        ref = oracle_warm(oracle, pb, g, True, max_iter=max_iter)
        gpu = run_warm(capi, pb, g, True, max_iter=max_iter)
        assert_bit_identical(gpu, ref, f"max_iter={max_iter}")
    ref = oracle_warm(oracle, pb, g, True)
    assert ref["term"][1] == 6 and ref["term"][2] == 2 and ref["term"][3] == 6
    assert ref["term"][4] == 0 and ref["term"][5] == 0


def test_warm_full_batch(capi, P, oracle):
    """BASELINE shape at the bench's batch size, 4096 x (n=50, m=200), with stale guesses."""
    pb = P.generate_batch(4096, P.benchmark_characteristics(50, 200))
    cold = oracle_cold(oracle, pb, False)
    g = perturbed_guess(np.random.default_rng(11), cold["status"], pb.m, flip=0.2, add=0.02)
    ref = oracle_warm(oracle, pb, g, False)
    gpu = run_warm(capi, pb, g)
    assert_bit_identical(gpu, ref, "4096 warm, stale guesses")
    assert np.all(gpu["term"] == 0)
    assert np.max(np.linalg.norm(gpu["x"] - pb.x_star, axis=1)) <= 1e-6
