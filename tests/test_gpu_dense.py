"""GPU parity tests of Mode A (one dense QP, state in HBM, host-driven hooks: jrlqp_dense_* of the C ABI) against
the CPU oracle, bit for bit: termination status, iterations, active set, x, u, f and the Cholesky factor left in G
(blocked LLT with the MFMA trailing update)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


@pytest.fixture(scope="module")
def P(pkg):
    import jrl_qp_amd.problems as problems

    return problems


def solve_gpu(capi, pb, b, use_bounds, max_iter=500):
    import torch

    n, m = pb.n, pb.m
    G = torch.from_numpy(np.ascontiguousarray(pb.G[b])).cuda()
    a = torch.from_numpy(np.ascontiguousarray(pb.a[b])).cuda()
    C = torch.from_numpy(np.ascontiguousarray(pb.C[b])).cuda()
    bl = torch.from_numpy(np.ascontiguousarray(pb.bl[b])).cuda()
    bu = torch.from_numpy(np.ascontiguousarray(pb.bu[b])).cuda()
    xl = torch.from_numpy(np.ascontiguousarray(pb.xl[b])).cuda() if use_bounds else None
    xu = torch.from_numpy(np.ascontiguousarray(pb.xu[b])).cuda() if use_bounds else None
    s = capi.DenseSolver(n, m, use_bounds, max_iter=max_iter)
    r = s.solve(G, a, C, bl, bu, xl, xu)
    torch.cuda.synchronize()
    r["L"] = G.cpu().numpy().T  # column-major view
    return r


def check(gpu, ref, what):
    assert gpu["term"] == ref.term, what
    assert gpu["iters"] == ref.iters, what
    assert np.array_equal(gpu["status"], ref.status), what
    assert np.array_equal(gpu["x"], ref.x), f"{what}: x max diff {np.max(np.abs(gpu['x'] - ref.x)):.3e}"
    assert np.array_equal(gpu["u"], ref.u), what
    assert gpu["f"] == ref.f, what


CASES = [
    dict(nVar=10, nObj=10, nIneq=20, nStrongActIneq=3),  # BASELINE c1
    dict(nVar=50, nObj=50, nIneq=200, nStrongActIneq=15),  # c3 shape through the host-driven path
    dict(nVar=33, nObj=33, nEq=5, nIneq=70, nStrongActIneq=10, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),
    dict(nVar=100, nObj=100, nIneq=260, nStrongActIneq=30),  # several LLT panels, partial MFMA tiles (100 = 3*32 + 4)
    dict(nVar=129, nObj=129, nEq=7, nIneq=150, nStrongActIneq=40, bounds=True, nStrongActBounds=9),
    dict(nVar=5, nObj=5),
    dict(nVar=1, nObj=1, nIneq=3, nStrongActIneq=1),
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_dense_path_matches_oracle(capi, P, oracle, ci):
    ch = P.Characteristics(**CASES[ci])
    pb = P.generate_batch(3, ch, first_index=900 + 10 * ci)
    hb = ch.bounds
    for b in range(pb.B):
        ref = oracle.solve(pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], pb.xl[b] if hb else None, pb.xu[b] if hb else None)
        gpu = solve_gpu(capi, pb, b, hb)
        check(gpu, ref, f"case {ci} problem {b}")
        assert ref.term == 0
        n = pb.n
        assert np.array_equal(np.tril(gpu["L"]), np.tril(ref.L))  # factor left in G, lower triangle
        assert np.array_equal(np.triu(gpu["L"], 1), np.triu(pb.G[b], 1))  # strictly upper part untouched


def test_dense_n300(capi, P, oracle):
    """A mid-size problem: 10 LLT panels, hundreds of iterations, drops included."""
    ch = P.benchmark_characteristics(300, 600)
    pb = P.generate_batch(1, ch, first_index=4242)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=5000)
    gpu = solve_gpu(capi, pb, 0, False, max_iter=5000)
    check(gpu, ref, "n=300")
    assert ref.term == 0 and np.linalg.norm(gpu["x"] - pb.x_star[0]) <= 1e-6


def test_two_handles_on_two_threads(capi, P, oracle):
    """Two Mode A handles driven from two host threads at the same time, each on a stream of its own: every handle has its own pass
    tags, side streams and mailbox, so the solves must not see each other (ctypes releases the GIL inside jrlqp_dense_solve)."""
    import threading

    import torch

    chs = [P.benchmark_characteristics(260, 400), P.Characteristics(nVar=200, nObj=200, nEq=5, nIneq=300, nStrongActIneq=40, bounds=True, nStrongActBounds=15)]
    pbs = [P.generate_batch(1, ch, first_index=9100 + i) for i, ch in enumerate(chs)]
    refs = []
    for pb, ch in zip(pbs, chs):
        hb = ch.bounds
        refs.append(oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], pb.xl[0] if hb else None, pb.xu[0] if hb else None, max_iter=4000))
    out = [None, None]

    def work(i):
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(3):  # (repeated: the handle's tags keep counting across solves)
                out[i] = solve_gpu(capi, pbs[i], 0, chs[i].bounds, max_iter=4000)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert refs[i].term == 0
        check(out[i], refs[i], f"handle {i} beside the other")


@pytest.mark.parametrize("mode", ["events", "tags"])
def test_dense_stream_handoff_modes(capi, P, oracle, monkeypatch, mode):
    """The hand-off between the three streams of a pass: device-side tags (PassSync) or, where kernels of different streams do not run
    side by side (a counter-collecting profiler serialises the dispatches), events.  The handle probes at creation;
    JRLQP_DENSE_SYNC forces either.  Same bits both ways, bounds and general constraints, drops included."""
    monkeypatch.setenv("JRLQP_DENSE_SYNC", mode)
    ch = P.Characteristics(nVar=150, nObj=150, nEq=4, nIneq=260, nStrongActIneq=30, bounds=True, nStrongActBounds=12, doubleSidedIneq=True)
    pb = P.generate_batch(1, ch, first_index=8800)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], pb.xl[0], pb.xu[0], max_iter=3000)
    assert ref.term == 0 and ref.iters > 20
    check(solve_gpu(capi, pb, 0, True, max_iter=3000), ref, f"hand-off by {mode}")


def test_dense_c10_on_and_off(capi, P, oracle, monkeypatch):
    """CANON C10 (beyond 256 unknowns S = R^-1 is kept beside R and r = S d1 is a blocked product) against the literal form (r by back
    substitution on R) ON THE GPU: with JRLQP_DENSE_C10=0 the handle keeps no S and equals the oracle with ORC_LIT_R_SOLVE bit for bit;
    by default it equals the canonical oracle bit for bit; and the two agree in every decision (termination, iteration count,
    statuses) with x and u to rounding -- the same comparison tests/test_oracle_literal.py makes on 6942 problems on the CPU."""
    ch = P.Characteristics(nVar=300, nObj=300, nEq=6, nIneq=420, nStrongActIneq=80, bounds=True, nStrongActBounds=20, doubleSidedIneq=True)
    pb = P.generate_batch(1, ch, first_index=9100)
    canon = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], pb.xl[0], pb.xu[0], max_iter=5000)
    lit = oracle.solve_batch_literal(oracle.LIT_R_SOLVE, pb.G[:1], pb.a[:1], pb.C[:1], pb.bl[:1], pb.bu[:1], pb.xl[:1], pb.xu[:1], max_iter=5000)
    assert canon.term == 0 and canon.iters > 100
    on = solve_gpu(capi, pb, 0, True, max_iter=5000)
    check(on, canon, "C10 on")
    monkeypatch.setenv("JRLQP_DENSE_C10", "0")
    off = solve_gpu(capi, pb, 0, True, max_iter=5000)
    assert off["term"] == lit["term"][0] and off["iters"] == lit["iters"][0]
    assert np.array_equal(off["status"], lit["status"][0])
    assert np.array_equal(off["x"], lit["x"][0]) and np.array_equal(off["u"], lit["u"][0]) and off["f"] == lit["f"][0]
    # the decision: same verdict, same path, same set -- values to rounding
    assert off["term"] == on["term"] and off["iters"] == on["iters"] and np.array_equal(off["status"], on["status"])
    assert np.max(np.abs(off["x"] - on["x"])) <= 1e-11 * max(1.0, np.max(np.abs(on["x"])))


def test_c2_full_size(capi, P, oracle):
    """BASELINE configs[1] (c2) at full size: n = 1024, m = 2048, 307 strongly active, through jrlqp_dense_solve --
    termination status, iteration count, activation statuses, x, u, f and the Cholesky factor left in G, bit for bit
    against the oracle (32 LLT panels, ~500 iterations, drops included)."""
    n, m = 1024, 2048
    ch = P.benchmark_characteristics(n, m)
    assert ch.nStrongActIneq == 307
    pb = P.generate_batch(1, ch, first_index=0)  # the problem bench.py --workload c2 times on rank 0
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=100000)
    gpu = solve_gpu(capi, pb, 0, False, max_iter=100000)
    check(gpu, ref, "c2 full size")
    assert ref.term == 0 and ref.iters > 300
    assert np.array_equal(np.tril(gpu["L"]), np.tril(ref.L))
    assert np.array_equal(np.triu(gpu["L"], 1), np.triu(pb.G[0], 1))
    assert np.linalg.norm(gpu["x"] - pb.x_star[0]) <= 1e-6 * max(1.0, np.linalg.norm(pb.x_star[0]))
    assert int((gpu["status"] != 0).sum()) >= 307


def test_dense_beyond_1024(capi, P, oracle):
    """n = 1400 with 1300 strongly active constraints: more than 1024 columns of R (remove_r_kernel<4>: four columns per
    thread; 28 of the 183 removals happen there), sweeps of more than 1024 rotations, LDS beyond 64 KiB in rot_apply."""
    ch = P.Characteristics(nVar=1400, nObj=1400, nIneq=1800, nStrongActIneq=1300)
    pb = P.generate_batch(1, ch, first_index=77)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=100000, trace_cap=4000)
    q, deep = 0, 0
    for (_, _, _, kind, _, _) in ref.trace:
        if kind == 0:
            q += 1
        else:
            deep += q > 1024
            q -= 1
    assert ref.term == 0 and deep >= 10
    gpu = solve_gpu(capi, pb, 0, False, max_iter=100000)
    check(gpu, ref, "n=1400")
    assert np.array_equal(np.tril(gpu["L"]), np.tril(ref.L))


def test_dense_n2800(capi, P, oracle):
    """Beyond 2730 unknowns the step kernel's three staged vectors need more than the default 64 KiB of LDS, the sweep's
    (c, s) table too, and the coefficient chain its third vector (all raised to the CU's 160 KiB once)."""
    ch = P.Characteristics(nVar=2800, nObj=2800, nIneq=300, nStrongActIneq=60)
    pb = P.generate_batch(1, ch, first_index=5)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=100000)
    gpu = solve_gpu(capi, pb, 0, False, max_iter=100000)
    check(gpu, ref, "n=2800")
    assert ref.term == 0 and ref.iters >= 60
    assert np.array_equal(np.tril(gpu["L"]), np.tril(ref.L))


def test_dense_at_the_size_limit(capi, P, oracle):
    """n = JRLQP_DENSE_MAX_N: the sweep's LDS at the CU's 160 KiB, 64 blocks in the coefficient scan, `jtn` / the scan with more
    workgroups than CUs -- and the stream hand-off by events (beyond 2048 unknowns the spinning consumers of the tag hand-off could
    starve their producers of LDS: tools/check_max_size.py found a solve at n = 3900 stalling until the wait gave up)."""
    n = 4096
    pb = P.generate_batch(1, P.benchmark_characteristics(n, 96, 30), first_index=123)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=80)
    gpu = solve_gpu(capi, pb, 0, False, max_iter=80)
    check(gpu, ref, "n = 4096")
    assert ref.term == 0 and ref.iters > 30


def test_dense_size_limit(capi):
    with pytest.raises(capi.JrlqpError):
        capi.DenseSolver(capi.DENSE_MAX_N + 1, 1, False)


def test_dense_status_codes(capi, P, oracle):
    n, m = 4, 6
    rng = np.random.default_rng(5)
    G = np.eye(n)
    a = rng.standard_normal(n)
    C = rng.standard_normal((m, n))
    bl = np.full(m, -np.inf)
    bu = np.full(m, -5.0)
    mk = lambda G_, bl_, bu_: P.PackedBatch(n=n, m=m, nb=0, G=G_[None], a=a[None], C=C[None], bl=bl_[None], bu=bu_[None], xl=np.zeros((1, 0)),
                                            xu=np.zeros((1, 0)), x_star=np.zeros((1, n)), lam_star=np.zeros((1, m)))
    # MAX_ITER_REACHED
    pb = mk(G, bl, bu)
    ref = oracle.solve(G, a, C.T, bl, bu, max_iter=2)
    check(solve_gpu(capi, pb, 0, False, max_iter=2), ref, "max iter")
    # NON_POS_HESSIAN at column 2 and at column 0
    for col in (2, 0):
        Gb = np.eye(n)
        Gb[col, col] = -1.0
        ref = oracle.solve(Gb, a, C.T, bl, bu)
        g = solve_gpu(capi, mk(Gb, bl, bu), 0, False)
        assert g["term"] == ref.term == 2
    # INFEASIBLE
    C2 = C.copy()
    C2[0] = [1, 0, 0, 0]
    C2[1] = [-1, 0, 0, 0]
    bu2 = np.full(m, 10.0)
    bu2[0] = bu2[1] = -1.0
    pb2 = P.PackedBatch(n=n, m=m, nb=0, G=G[None], a=a[None], C=C2[None], bl=bl[None], bu=bu2[None], xl=np.zeros((1, 0)), xu=np.zeros((1, 0)),
                        x_star=np.zeros((1, n)), lam_star=np.zeros((1, m)))
    ref = oracle.solve(G, a, C2.T, bl, bu2)
    check(solve_gpu(capi, pb2, 0, False), ref, "infeasible")
    assert ref.term == 3


def solve_gpu_warm(capi, pb, b, use_bounds, guess, max_iter=500):
    import torch

    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    s = capi.DenseSolver(pb.n, pb.m, use_bounds, max_iter=max_iter)
    r = s.solve(dev(pb.G[b]), dev(pb.a[b]), dev(pb.C[b]), dev(pb.bl[b]), dev(pb.bu[b]), dev(pb.xl[b]) if use_bounds else None,
                dev(pb.xu[b]) if use_bounds else None, warm=True, guess=guess)
    torch.cuda.synchronize()
    return r


DENSE_WARM = [
    dict(nVar=20, nObj=20, nIneq=40, nStrongActIneq=6),
    dict(nVar=33, nObj=33, nEq=5, nIneq=70, nStrongActIneq=10, bounds=True, nStrongActBounds=4, doubleSidedIneq=True),
    dict(nVar=129, nObj=129, nEq=7, nIneq=150, nStrongActIneq=40, bounds=True, nStrongActBounds=9),
    dict(nVar=300, nObj=300, nIneq=400, nStrongActIneq=90),
    dict(nVar=300, nObj=300, nEq=20, nIneq=260, nStrongActIneq=60, bounds=True, nStrongActBounds=25, doubleSidedIneq=True),
]


@pytest.mark.parametrize("ci", range(len(DENSE_WARM)))
def test_dense_warm_start_matches_oracle(capi, P, oracle, ci):
    """jrlqp_dense_solve_warm (experimental::GoldfarbIdnaniSolver with warmStart_, src/experimental/GoldfarbIdnaniSolver.cpp:20-112,
    306-487) == the oracle's orc_gi_solve_warm, bit for bit: no guess, the exact active set (0 iterations -- the reference's own
    warm-start expectation, tests/GoldfarbIdnaniSolverTest.cpp:176,181), stale guesses (purge loop + regular loop) and an
    all-lower guess (more than n guesses: trimmed from the back)."""
    from test_gpu_warm import perturbed_guess

    ch = P.Characteristics(**DENSE_WARM[ci])
    pb = P.generate_batch(2, ch, first_index=7700 + 10 * ci, seed0=0xD3A5)
    hb = ch.bounds
    L, LB = 1, 4
    purges = 0
    for b in range(pb.B):
        args = (pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], pb.xl[b] if hb else None, pb.xu[b] if hb else None)
        cold = oracle.solve(*args, max_iter=5000)
        rng = np.random.default_rng(10 * ci + b)
        st2 = cold.status[None, :]
        guesses = {
            "none": None,
            "exact": cold.status,
            "stale": perturbed_guess(rng, st2, pb.m)[0],
            "very stale": perturbed_guess(rng, st2, pb.m, flip=0.5, add=0.2)[0],
            "everything lower": np.where(np.arange(pb.m + pb.nb) < pb.m, L, LB).astype(np.int32),
        }
        for name, g in guesses.items():
            ref = oracle.solve(*args, max_iter=5000, warm=True, guess=g)
            gpu = solve_gpu_warm(capi, pb, b, hb, g, max_iter=5000)
            check(gpu, ref, f"{DENSE_WARM[ci]} problem {b} guess '{name}'")
            if name == "exact" and ch.nEq == 0:
                assert ref.term == 0 and ref.iters == 0
            purges += ref.iters


def test_dense_warm_error_paths(capi, P, oracle):
    """OVERCONSTRAINED before the factorisation, NON_POS_HESSIAN, the iteration limit inside / behind the purge loop."""
    from test_gpu_warm import perturbed_guess

    ch = P.Characteristics(nVar=40, nObj=40, nEq=3, nIneq=60, nStrongActIneq=8, bounds=True, nStrongActBounds=3)
    pb = P.generate_batch(4, ch, first_index=91, seed0=0xE46)
    pb.bu[1, :41] = pb.bl[1, :41]
    pb.G[2, 0, 0] = -1.0
    pb.bu[3, :41] = pb.bl[3, :41]
    pb.G[3, 0, 0] = -1.0
    for b in range(4):
        args = (pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], pb.xl[b], pb.xu[b])
        cold = oracle.solve(*args)
        g = perturbed_guess(np.random.default_rng(b), cold.status[None, :], pb.m, flip=0.5, add=0.3)[0]
        for max_iter in (500, 2, 0):
            ref = oracle.solve(*args, max_iter=max_iter, warm=True, guess=g)
            gpu = solve_gpu_warm(capi, pb, b, True, g, max_iter=max_iter)
            check(gpu, ref, f"problem {b} max_iter={max_iter}")
    assert oracle.solve(pb.G[1], pb.a[1], pb.C[1].T, pb.bl[1], pb.bu[1], pb.xl[1], pb.xu[1], warm=True).term == 6


def _torch_problem(pb, b, use_bounds):
    import torch

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return (t(pb.G[b]), t(pb.a[b]), t(pb.C[b]), t(pb.bl[b]), t(pb.bu[b]), t(pb.xl[b]) if use_bounds else None, t(pb.xu[b]) if use_bounds else None)


def test_dense_trace_matches_oracle(capi, P, oracle):
    """jrlqp_dense_set_trace: one record per pass of Mode A's loop -- (p, status, l, kind, t1, t2), the counterpart of the reference's
    per-pass log lines (DualSolver.cpp:105-159) -- equal to the oracle's trace, field by field, bit by bit."""
    ch = P.Characteristics(nVar=60, nObj=60, nEq=4, nIneq=120, nStrongActIneq=18, bounds=True, nStrongActBounds=5, doubleSidedIneq=True)
    pb = P.generate_batch(2, ch, first_index=5150)
    for b in range(pb.B):
        ref = oracle.solve(pb.G[b], pb.a[b], pb.C[b].T, pb.bl[b], pb.bu[b], pb.xl[b], pb.xu[b], trace_cap=400)
        s = capi.DenseSolver(pb.n, pb.m, True)
        s.set_trace(400)
        r = s.solve(*_torch_problem(pb, b, True))
        check(r, ref, f"problem {b}")
        got = s.trace_numpy(r["iters"])
        assert len(got) == ref.iters == len(ref.trace) and ref.iters > 10
        assert got == ref.trace
        assert {k for (_, _, _, k, _, _) in got} >= {0, 1}  # additions and drops both seen
        s.set_trace(0)  # switched off again: the next solve writes nothing
        r2 = s.solve(*_torch_problem(pb, b, True))
        check(r2, ref, f"problem {b}, trace off")


def test_dense_vector_trace_matches_oracle(capi, P, oracle):
    """jrlqp_dense_set_trace_vectors (round 5): x, u, f at the top of every pass and z, r behind its step computation -- what the
    reference logs under LogFlags::ITERATION_BASIC_DETAILS (DualSolver.cpp:108, :128-129) -- bit for bit the oracle's, pass by pass; and
    the solve's results are those of an untraced solve (the trace only synchronises)."""
    import jrl_qp_amd.trace as T

    ch = P.Characteristics(nVar=60, nObj=60, nEq=4, nIneq=120, nStrongActIneq=18, bounds=True, nStrongActBounds=5, doubleSidedIneq=True)
    pb = P.generate_batch(1, ch, first_index=5150)
    ref, vec = oracle.solve_with_vectors(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], pb.xl[0], pb.xu[0], cap=400, trace_cap=400)
    s = capi.DenseSolver(pb.n, pb.m, True)
    s.set_trace(400)
    s.set_trace_vectors(400)
    r = s.solve(*_torch_problem(pb, 0, True))
    check(r, ref, "vector trace on")
    got = s.trace_vectors(r["iters"])
    assert len(got) == len(vec) == ref.iters > 10
    for it, (g, o) in enumerate(zip(got, vec)):
        assert g["q"] == o["q"] and g["f"] == o["f"], it
        for k in ("x", "u", "z", "r"):
            assert np.array_equal(g[k], o[k]), (it, k)
    txt = T.to_matlab(s.trace_numpy(r["iters"]), name="gi", vectors=got)
    assert "gi(1).x = [" in txt and "gi(3).z = [" in txt and "gi(3).r = [" in txt and txt.count(".f = ") == ref.iters
    s.set_trace_vectors(0)
    check(s.solve(*_torch_problem(pb, 0, True)), ref, "vector trace off")
    # beyond 256 unknowns (the blocked scan that otherwise runs ahead of the decision)
    ch = P.Characteristics(nVar=300, nObj=300, nIneq=330, nStrongActIneq=40)
    pb = P.generate_batch(1, ch, first_index=77)
    ref, vec = oracle.solve_with_vectors(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], cap=8)
    s = capi.DenseSolver(pb.n, pb.m, False)
    s.set_trace_vectors(8)
    r = s.solve(*_torch_problem(pb, 0, False))
    check(r, ref, "vector trace, n = 300")
    got = s.trace_vectors(r["iters"])
    assert len(got) == 8
    for it, (g, o) in enumerate(zip(got, vec)):
        for k in ("x", "u", "z", "r"):
            assert np.array_equal(g[k], o[k]), (it, k)


def test_dense_no_constraint_twice_on_one_handle(capi, P, oracle):
    """m = 0 and nb = 0 beyond the batched sizes: nothing is selected, so no kernel of a pass may be left waiting for a tag nobody
    posts (round 3's review: the side-stream chain and z spun for a second and poisoned the next solve).  Two solves in a row."""
    import time

    pb = P.generate_batch(2, P.Characteristics(nVar=150, nObj=150), first_index=5170)
    s = capi.DenseSolver(pb.n, 0, False)
    for b in range(2):
        ref = oracle.solve(pb.G[b], pb.a[b], np.zeros((pb.n, 0)), np.zeros(0), np.zeros(0))
        t0 = time.perf_counter()
        r = s.solve(*_torch_problem(pb, b, False))
        dt = time.perf_counter() - t0
        check(r, ref, f"solve {b}")
        assert r["term"] == 0 and r["iters"] == 0
        assert dt < 0.5, f"solve {b} took {dt:.2f} s: something waited for a tag"
    assert s.sync_mode()[1] == 0


@pytest.mark.timeout(600)
def test_dense_starved_handoff_is_rerun_with_events(capi, P, oracle, monkeypatch):
    """A device-side wait between the streams that gives up (fault injection: pass 3's first kernel waits for a tag nobody posts,
    ~1 s) must not surface as an error: the handle drains its streams, switches to event edges for good and repeats the solve from
    the factor it already has -- same bits as the oracle, G holds L as always, and the next solve on the handle is clean."""
    ch = P.Characteristics(nVar=140, nObj=140, nIneq=200, nStrongActIneq=30)
    pb = P.generate_batch(2, ch, first_index=5190)
    # (tags whatever the creation-time probe says: when the handle's two side streams happen to share a hardware queue the probe
    # -- which enqueues a consumer BEFORE its producer on purpose -- picks events, and there would be nothing to starve)
    monkeypatch.setenv("JRLQP_DENSE_SYNC", "tags")
    s = capi.DenseSolver(pb.n, pb.m, False)
    monkeypatch.delenv("JRLQP_DENSE_SYNC")
    assert s.sync_mode() == (0, 0)
    s.inject_sync_fault(3)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0])
    args = _torch_problem(pb, 0, False)
    r = s.solve(*args)
    check(r, ref, "rerun")
    assert ref.iters > 5
    assert s.sync_mode() == (1, 1)
    L = args[0].cpu().numpy().T
    assert np.array_equal(np.tril(L), np.tril(ref.L)) and np.array_equal(np.triu(L, 1), np.triu(pb.G[0], 1))
    s.inject_sync_fault(-1)
    ref1 = oracle.solve(pb.G[1], pb.a[1], pb.C[1].T, pb.bl[1], pb.bu[1])
    check(s.solve(*_torch_problem(pb, 1, False)), ref1, "next solve, event mode")
    assert s.sync_mode() == (1, 1)


@pytest.mark.timeout(600)
def test_dense_with_a_third_party_on_the_cus(capi, P, oracle):
    """Contention: another stream of the process keeps every CU busy with large fp64 products while a handle solves with the tag
    hand-off.  The solve must come back with the oracle's bits -- through the tags or, if they starved, through the event rerun --
    and never hang (pytest's per-test timeout and the 1 s give-up of every device-side wait bound it)."""
    import threading
    import time

    import torch

    ch = P.Characteristics(nVar=200, nObj=200, nIneq=300, nStrongActIneq=50)
    pb = P.generate_batch(1, ch, first_index=5210)
    ref = oracle.solve(pb.G[0], pb.a[0], pb.C[0].T, pb.bl[0], pb.bu[0], max_iter=2000)
    stop = threading.Event()
    launched = [0]

    def hog():
        torch.cuda.set_device(0)
        side = torch.cuda.Stream()
        A = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(4):
                    A = torch.mm(A, A)
                    A = A / A.abs().max()
                launched[0] += 4
                side.synchronize()

    th = threading.Thread(target=hog)
    th.start()
    try:
        time.sleep(0.3)  # the products are running
        s = capi.DenseSolver(pb.n, pb.m, False, max_iter=2000)
        t0 = time.perf_counter()
        r = s.solve(*_torch_problem(pb, 0, False))
        dt = time.perf_counter() - t0
    finally:
        stop.set()
        th.join()
    check(r, ref, "under contention")
    assert launched[0] > 0
    mode, reruns = s.sync_mode()
    print(f"contention: {ref.iters} passes in {dt * 1e3:.0f} ms beside {launched[0]} 4096^3 fp64 products; sync mode {mode}, event reruns {reruns}")
