"""Pins the CPU oracle (oracle/jrlqp_oracle.c) to the reference's own known answers and fixtures
(SURVEY.md §8c): GI-paper example, "Simple problem", planted-solution "Random problems", MultiIK fixtures.
CPU only."""
import numpy as np
import pytest

from ikdata import read_arrow, read_tri_block_diag

I, L, U, E, LB, UB, F = range(7)


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


@pytest.fixture(scope="module")
def K(pkg):
    import jrl_qp_amd.kkt as kkt

    return kkt


def test_paper_example(oracle, K):
    """reference tests/GoldfarbIdnaniSolverTest.cpp:51-73; analytic optimum x*=(1/2,3/2), f*=6.5,
    multipliers (-5,0,0), active set (LOWER, INACTIVE, INACTIVE), 1 iteration."""
    G = np.array([[4.0, -2.0], [-2.0, 4.0]])
    a = np.array([6.0, 0.0])
    C = np.array([[1.0], [1.0]])
    r = oracle.solve(G, a, C, [2.0], [10.0], [0.0, 0.0], [10.0, 10.0], trace_cap=4)
    assert r.term == 0
    assert np.allclose(r.x, [0.5, 1.5], rtol=0, atol=1e-14)
    assert abs(r.f - 6.5) < 1e-14
    assert np.allclose(r.u, [-5.0, 0.0, 0.0], rtol=0, atol=1e-13)
    assert list(r.status) == [L, I, I]
    assert list(r.active) == [0]
    assert r.iters == 1
    assert K.test_kkt(r.x, r.u, G, a, C, np.array([2.0]), np.array([10.0]), np.zeros(2), np.full(2, 10.0))


def test_simple_problem(oracle, K):
    """reference tests/GoldfarbIdnaniSolverTest.cpp:23-49 (G = I, a = 0, random 3x5 C, |C^T x| <= 1)."""
    rng = np.random.default_rng(7)
    G = np.eye(3)
    a = np.zeros(3)
    C = rng.uniform(-1, 1, (3, 5))
    bl = -np.ones(5)
    bu = np.ones(5)
    r = oracle.solve(G, a, C, bl, bu)
    assert r.term == 0 and r.iters == 0 and np.all(r.x == 0)
    assert K.test_kkt(r.x, r.u, G, a, C, bl, bu)
    bl[1], bu[1] = -2.0, -1.0
    r = oracle.solve(G, a, C, bl, bu)
    assert r.term == 0 and r.iters >= 1
    assert K.test_kkt(r.x, r.u, G, a, C, bl, bu)
    # closed form: projection of 0 on {c1^T x <= -1} if no other constraint is hit
    c1 = C[:, 1]
    if r.iters == 1:
        assert np.allclose(r.x, -c1 / (c1 @ c1), atol=1e-14)


CHARACS = [
    dict(nVar=5, nObj=5),
    dict(nVar=5, nObj=5, nEq=2),
    dict(nVar=5, nObj=5, nIneq=8, nStrongActIneq=4),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=3),
    dict(nVar=5, nObj=5, nEq=2, nIneq=6, nStrongActIneq=1, bounds=True, nStrongActBounds=2),
    # beyond the reference's list: double-sided, bounds only, everything at once
    dict(nVar=6, nObj=6, nIneq=10, nStrongActIneq=3, doubleSidedIneq=True),
    dict(nVar=8, nObj=8, bounds=True, nStrongActBounds=3),
    dict(nVar=10, nObj=10, nEq=3, nIneq=20, nStrongActIneq=3, bounds=True, nStrongActBounds=2, doubleSidedIneq=True),
]


@pytest.mark.parametrize("ci", range(len(CHARACS)))
def test_random_problems_planted(oracle, P, K, ci):
    """reference tests/GoldfarbIdnaniSolverTest.cpp:75-99: SUCCESS, KKT, x ~ x*, multipliers ~ planted (rel 1e-6)."""
    ch = P.Characteristics(**CHARACS[ci])
    for rep in range(20):
        pb = P.random_problem(ch, P.make_rng(1000 * ci + rep))
        G, a, C, bl, bu, xl, xu = pb.to_qp()
        has_b = ch.bounds
        r = oracle.solve(G, a, C, bl, bu, xl if has_b else None, xu if has_b else None)
        assert r.term == 0
        assert K.test_kkt(r.x, r.u, G, a, C, bl, bu, xl if has_b else None, xu if has_b else None)
        assert np.linalg.norm(r.x - pb.x) <= 1e-6 * max(np.linalg.norm(pb.x), 1e-300) + 1e-12
        lam = pb.multipliers()
        if ch.nWeakActIneq == 0 and ch.nWeakActBounds == 0:
            assert np.linalg.norm(r.u - lam) <= 1e-6 * max(np.linalg.norm(lam), 1.0)
        # equalities are activated as EQUALITY, in index order, before anything else (GoldfarbIdnaniSolver.cpp:268-287)
        assert list(r.status[: ch.nEq]) == [E] * ch.nEq
        assert list(r.active[: ch.nEq]) == list(range(ch.nEq))


GEN_CHARACS = CHARACS + [
    dict(nVar=6, nObj=6, nIneq=10, nStrongActIneq=2, nWeakActIneq=2),
    dict(nVar=8, nObj=8, bounds=True, nStrongActBounds=3, nWeakActBounds=1),
    dict(nVar=7, nObj=4, nEq=1, nIneq=6, nStrongActIneq=1),
    dict(nVar=7, nObj=7, rankObj=5, nEq=1, nIneq=6, nStrongActIneq=2),
]


@pytest.mark.parametrize("ci", range(len(GEN_CHARACS)))
def test_generator_planted_point_satisfies_kkt(P, K, ci):
    """reference tests/RandomProblemsTest.cpp:122-140: the generated (x, lambda) satisfies testKKT."""
    ch = P.Characteristics(**GEN_CHARACS[ci])
    for rep in range(10):
        pb = P.random_problem(ch, P.make_rng(5000 + 100 * ci + rep))
        G, a, C, bl, bu, xl, xu = pb.to_qp()
        hb = ch.bounds
        assert K.test_kkt(pb.x, pb.multipliers(), G, a, C, bl, bu, xl if hb else None, xu if hb else None)


def test_benchmark_recipe_c1_c3(oracle, P, K):
    """The BASELINE.json workloads (c1: n=10,m=20,ma=3; c3: n=50,m=200,ma=15) solve to the planted optimum."""
    for n, m, cnt in ((10, 20, 50), (50, 200, 10)):
        ch = P.benchmark_characteristics(n, m)
        pb = P.generate_batch(cnt, ch)
        res = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu)
        assert np.all(res["term"] == 0)
        assert np.max(np.linalg.norm(res["x"] - pb.x_star, axis=1)) <= 1e-6
        assert np.max(np.abs(res["u"] - pb.lam_star)) <= 1e-6
        for b in range(cnt):
            assert K.test_kkt(res["x"][b], res["u"][b], pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b])
            # the optimal active set is the planted one
            assert np.array_equal(res["status"][b] == U, pb.lam_star[b] > 0)


def test_batch_equals_single_and_threads(oracle, P):
    ch = P.benchmark_characteristics(10, 20)
    pb = P.generate_batch(16, ch, first_index=77)
    r1 = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=1)
    r4 = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu, nthreads=4)
    for k in ("x", "u", "status", "term", "iters", "f"):
        assert np.array_equal(r1[k], r4[k])
    for b in range(16):
        r = oracle.solve(pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b])
        assert np.array_equal(r.x, r1["x"][b]) and r.iters == r1["iters"][b] and r.f == r1["f"][b]


def test_status_codes(oracle):
    # NON_POS_HESSIAN (GoldfarbIdnaniSolver.cpp:61), including a failure at column 0 (ret >= 0)
    r = oracle.solve(np.array([[-1.0, 0], [0, 1.0]]), np.zeros(2), np.zeros((2, 0)), [], [])
    assert r.term == 2
    r = oracle.solve(np.array([[1.0, 2.0], [2.0, 1.0]]), np.zeros(2), np.zeros((2, 0)), [], [])
    assert r.term == 2
    # INFEASIBLE (DualSolver.cpp:131): x >= 1 and x <= 0 in 1-D
    r = oracle.solve(np.eye(1), np.zeros(1), np.array([[1.0, 1.0]]), [1.0, -np.inf], [np.inf, 0.0])
    assert r.term == 3
    # MAX_ITER_REACHED (DualSolver.cpp:167)
    G = np.eye(3)
    C = np.eye(3)
    r = oracle.solve(G, np.zeros(3), C, [1.0, 1.0, 1.0], [2.0, 2.0, 2.0], max_iter=2)
    assert r.term == 4 and r.iters == 2
    r = oracle.solve(G, np.zeros(3), C, [1.0, 1.0, 1.0], [2.0, 2.0, 2.0])
    assert r.term == 0 and r.iters == 3 and np.allclose(r.x, 1.0) and np.allclose(r.u, -1.0)
    # empty problem sizes: no constraints at all
    r = oracle.solve(np.array([[4.0]]), np.array([-8.0]), np.zeros((1, 0)), [], [])
    assert r.term == 0 and r.x[0] == 2.0 and r.f == -8.0 and r.iters == 0


def test_drop_path_is_exercised(oracle, P):
    """At least one problem of the benchmark set takes partial steps (drops) -- Q-U / K11 are covered."""
    ch = P.benchmark_characteristics(50, 200)
    pb = P.generate_batch(4, ch)
    kinds = set()
    for b in range(4):
        r = oracle.solve(pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], trace_cap=200)
        kinds |= {t[3] for t in r.trace}
        assert r.iters == len(r.trace)
        adds = sum(1 for t in r.trace if t[3] == 0)
        drops = sum(1 for t in r.trace if t[3] != 0)
        assert adds - drops == len(r.active)
    assert {0, 1} <= kinds


def test_multiik_tri_block_diag_fixture(oracle):
    """reference tests/BlockGISolverTest.in.cpp:172-188: SUCCESS and ||x - sol||_inf <= 1e-4."""
    d = read_tri_block_diag()
    assert d["G"].shape == (387, 387) and d["C"].shape == (387, 1621)
    r = oracle.solve(d["G"], d["a"], d["C"], d["bl"], d["bu"])
    assert r.term == 0
    assert np.max(np.abs(r.x - d["sol"])) <= 1e-4
    # SURVEY.md §8c probe of the same algorithm: 4 iterations, final active list [174, 171, 3, 0]
    assert r.iters == 4 and list(r.active) == [174, 171, 3, 0]


def test_multiik_arrow_fixture(oracle, K):
    """reference tests/BlockGISolverTest.in.cpp:273-286: dense GI returns SUCCESS on the arrow problem (box bounds)."""
    d = read_arrow()
    assert d["G"].shape == (210, 210) and d["C"].shape == (210, 25)
    r = oracle.solve(d["G"], d["a"], d["C"], d["bl"], d["bu"], d["xl"], d["xu"])
    assert r.term == 0
    assert K.test_kkt(r.x, r.u, d["G"], d["a"], d["C"], d["bl"], d["bu"], d["xl"], d["xu"])
    # SURVEY.md §8c probe: 5 iterations, active constraints are the bounds {124,220,208,94,166} (ids >= 25 are bounds)
    assert r.iters == 5 and sorted(r.active) == sorted([124, 220, 208, 94, 166])
