"""GPU parity of the structured GI solver (jrlqp_blockgi_*) against the CPU oracle (oracle/blockgi_oracle.c): termination status,
iteration count, activation-status vector AND x, u, f bit for bit (fp64 tolerance: 0 ulp).  The solver keeps the QR factors in one of
two forms (include/jrlqp_hip.h JRLQP_BLOCKGI_FORM_*): the reference's orthonormal sequence (blockgi.hip == orc_blockgi_solve) or contract
C9 (blockgi_c9.hip == orc_blockgi_solve_c9; the default for the arrows).  Every plan is compared with the oracle of the form it runs,
and both forms are driven on every shape (test_both_forms_*).  Cases: the reference's BlockGISolver tests (tests/BlockGISolverTest.in.cpp:70-351) -- small
tri-diagonal / arrow problems, the two MultiIK fixtures -- plus wider shapes, drops, bounds, and BASELINE config c5."""
import numpy as np
import pytest

from ikdata import read_arrow, read_tri_block_diag

pytestmark = pytest.mark.gpu

KIND = {"tri": "tridiag", "arrow_up": "arrow_up", "arrow_down": "arrow_down"}


@pytest.fixture(scope="module")
def SP(pkg):
    import jrl_qp_amd.structured_problems as sp

    return sp


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


FORM = {"form": None}  # the form the last GPU plan ran (0 sequence, 1 C9): the oracle to compare with


def gpu_solve(capi, qps, max_iter=500, trace_cap=0, form="auto", packed=False):
    """Solve a batch of identically shaped StructuredQP on the GPU."""
    import torch

    q0 = qps[0]
    B, n, m = len(qps), q0.n, q0.m
    use_bounds = q0.xl is not None
    shapes = [c.shape for c in q0.Cblocks]
    H = torch.tensor(np.stack([np.ascontiguousarray(q.G.T) for q in qps]), device="cuda")  # H[b].T = G
    Cb = torch.tensor(np.stack([np.concatenate([c.ravel(order="F") for c in q.Cblocks]) if m else np.zeros(1) for q in qps]), device="cuda")
    a = torch.tensor(np.stack([q.a for q in qps]), device="cuda")
    bl = torch.tensor(np.stack([q.bl for q in qps]), device="cuda")
    bu = torch.tensor(np.stack([q.bu for q in qps]), device="cuda")
    xl = torch.tensor(np.stack([q.xl for q in qps]), device="cuda") if use_bounds else None
    xu = torch.tensor(np.stack([q.xu for q in qps]), device="cuda") if use_bounds else None
    s = capi.BlockGISolver(KIND[q0.kind], q0.bs, shapes, use_bounds, B, max_iter=max_iter, form=form)
    FORM["form"] = s.form
    if packed:  # the blocks back to back, each with its own leading dimension, instead of views into the N x N matrices
        H = s.pack(H)
    r = s.solve(H, a, Cb, bl, bu, xl, xu, trace_cap=trace_cap)
    torch.cuda.synchronize()
    out = dict(x=r.x.cpu().numpy(), u=r.u.cpu().numpy()[:, : m + (n if use_bounds else 0)],
               status=r.status.cpu().numpy()[:, : m + (n if use_bounds else 0)], term=r.term.cpu().numpy(), iters=r.iters.cpu().numpy(),
               f=r.f.cpu().numpy(), L=H.cpu().numpy())
    return out


def oracle_solve(oracle, SP, qp, max_iter=500, c9=None):
    """The oracle of the form the GPU plan ran (c9=None: the last plan's)."""
    Gs = qp.G.copy(order="F")
    diag, off = SP.block_views(Gs, qp.bs, qp.kind)
    c9 = bool(FORM["form"]) if c9 is None else c9
    r = oracle.blockgi_solve(qp.kind, diag, off, qp.a, qp.Cblocks, qp.bl, qp.bu, qp.xl, qp.xu, max_iter=max_iter, c9=c9)
    return r, Gs


def check_batch(oracle, SP, capi, qps, max_iter=500, min_iters=0, form="auto", packed=False):
    g = gpu_solve(capi, qps, max_iter=max_iter, form=form, packed=packed)
    tot = 0
    for b, qp in enumerate(qps):
        r, Gs = oracle_solve(oracle, SP, qp, max_iter=max_iter)
        assert g["term"][b] == r.term, (b, g["term"][b], r.term)
        assert g["iters"][b] == r.iters, (b, g["iters"][b], r.iters)
        assert np.array_equal(g["status"][b], r.status), b
        assert np.array_equal(g["x"][b], r.x), (b, np.abs(g["x"][b] - r.x).max())
        assert np.array_equal(g["u"][b], r.u), b
        assert g["f"][b] == r.f, b
        tot += r.iters
    assert tot >= min_iters
    return g


@pytest.mark.parametrize("kind", ["tri", "arrow_up", "arrow_down"])
def test_small_problems_bit_exact(oracle, SP, capi, kind):
    """BlockGISolverTest.in.cpp:70-170: block sizes {3,5,2,3}, 3 double-sided constraints per block; a batch of 24 seeds."""
    qps = [SP.random_structured_qp(kind, [3, 5, 2, 3], [3, 3, 3, 3], seed) for seed in range(24)]
    check_batch(oracle, SP, capi, qps, min_iters=24)


@pytest.mark.parametrize("kind,bs,mc", [("tri", [7, 4, 9, 6, 5, 8], [4, 6, 3, 5, 7, 2]), ("arrow_up", [6, 9, 4, 7, 5], [5, 3, 6, 4, 2]),
                                        ("arrow_down", [8, 5, 7, 6, 9], [3, 6, 4, 5, 0])])
def test_wider_problems_bit_exact(oracle, SP, capi, kind, bs, mc):
    qps = [SP.random_structured_qp(kind, bs, mc, 100 + seed, ridge=0.05) for seed in range(16)]
    check_batch(oracle, SP, capi, qps, min_iters=100)


@pytest.mark.parametrize("kind", ["tri", "arrow_up", "arrow_down"])
def test_unaligned_c_blocks_and_box_bounds_bit_exact(oracle, SP, capi, kind):
    """C blocks that do not follow G's partition (windows straddle blocks of G) + bound constraints."""
    qps = [SP.random_structured_qp(kind, [6, 9, 4, 7, 5], [5, 7, 6], 300 + seed, ridge=0.05, cbs=[10, 8, 13], box=1.0) for seed in range(12)]
    g = check_batch(oracle, SP, capi, qps, min_iters=60)
    assert (g["status"][:, qps[0].m :] != 0).any()


def test_sequential_ik_fixture(oracle, SP, capi):
    """BlockGISolverTest.in.cpp:172-227 (tri-block 9 x 43, m = 1621)."""
    d = read_tri_block_diag()
    nd, C = d["block"], d["C"]
    first = np.array([int(np.flatnonzero(C[:, i])[0]) for i in range(C.shape[1])])
    blk = first // nd
    Cb = [np.asfortranarray(C[i * nd : (i + 1) * nd, blk == i]) for i in range(d["nblk"])]
    qp = SP.StructuredQP("tri", [nd] * d["nblk"], np.asfortranarray(d["G"]), d["a"], Cb, d["bl"], d["bu"])
    g = check_batch(oracle, SP, capi, [qp, qp])
    assert g["term"][0] == 0 and g["iters"][0] == 4
    assert np.abs(g["x"][0] - d["sol"]).max() <= 1e-4


def test_simultaneous_ik_fixture(oracle, SP, capi):
    """BlockGISolverTest.in.cpp:273-307 (arrow up 5 x 42, 25 inequalities + box bounds)."""
    d = read_arrow()
    nd, k, C = d["block"], d["nblk"], d["C"]
    Cb = [np.asfortranarray(C[i * nd : (i + 1) * nd, 5 * i : 5 * (i + 1)]) for i in range(k)]
    qp = SP.StructuredQP("arrow_up", [nd] * k, np.asfortranarray(d["G"]), d["a"], Cb, d["bl"], d["bu"], d["xl"], d["xu"])
    g = check_batch(oracle, SP, capi, [qp])
    assert g["term"][0] == 0 and g["iters"][0] == 5


def test_equalities_and_indefinite_are_reported(oracle, SP, capi):
    qa = SP.random_structured_qp("tri", [3, 5, 2, 3], [3, 3, 3, 3], 3)
    qb = SP.random_structured_qp("tri", [3, 5, 2, 3], [3, 3, 3, 3], 4)
    qb.bl[4] = qb.bu[4]
    qc = SP.random_structured_qp("tri", [3, 5, 2, 3], [3, 3, 3, 3], 5)
    qc.G[5, 5] = -1.0
    g = check_batch(oracle, SP, capi, [qa, qb, qc])
    assert list(g["term"]) == [0, 1, 2]


def test_mpc_arrow_medium(oracle, SP, capi):
    qps = [SP.mpc_arrow_qp(nstage=12, nx=10, mc=4, seed=s) for s in range(8)]
    check_batch(oracle, SP, capi, qps, min_iters=40)


@pytest.mark.parametrize("kind", ["tri", "arrow_up", "arrow_down"])
def test_vertex_solutions_bit_exact(oracle, SP, capi, kind):
    """Many binding constraints: the active set fills up to q == n (reflectors of length one, empty J2, dual-only steps)."""
    qps = []
    for seed in range(40):
        qp = SP.random_structured_qp(kind, [2, 1, 2], [4, 3, 4], 900 + seed, ridge=0.05, box=0.3)
        qp.a *= 30  # the unconstrained optimum is far outside the box
        qps.append(qp)
    g = check_batch(oracle, SP, capi, qps)
    assert ((g["status"] != 0).sum(axis=1) == qps[0].n).sum() >= 3  # some solutions are vertices


def test_degenerate_shapes_bit_exact(oracle, SP, capi):
    """One diagonal block only (the block list degenerates: arrow = head only, tridiagonal = one block), bounds without
    general constraints, and no constraint at all (x = -G^-1 a, zero iterations)."""
    for kind in ("tri", "arrow_up", "arrow_down"):
        qps = [SP.random_structured_qp(kind, [7], [9], 40 + s, ridge=0.05, box=0.5) for s in range(4)]
        check_batch(oracle, SP, capi, qps, min_iters=4)
        # bounds only
        qps = [SP.random_structured_qp(kind, [4, 6, 3], [0, 0, 0], 60 + s, ridge=0.05, box=0.05) for s in range(4)]
        g = check_batch(oracle, SP, capi, qps, min_iters=4)
        # unconstrained
        qps = [SP.random_structured_qp(kind, [4, 6, 3], [0, 0, 0], 80 + s, ridge=0.05) for s in range(2)]
        g = check_batch(oracle, SP, capi, qps)
        assert (g["iters"] == 0).all() and (g["term"] == 0).all()


def test_throughput_layout_large_batch(oracle, SP, capi):
    """n > 1024 in a batch of >= 128 problems selects the throughput layout (x and u in HBM, four waves, two problems per
    CU): same bits as the latency layout and as the oracle, for every problem of the batch."""
    base = [SP.mpc_arrow_qp(nstage=36, nx=30, mc=4, seed=s, scale_a=1.0) for s in range(2)]
    assert base[0].n > 1024
    qps = [base[i % 2] for i in range(128)]
    g = gpu_solve(capi, qps, max_iter=600)
    refs = [oracle_solve(oracle, SP, q, max_iter=600)[0] for q in base]
    for b in range(128):
        r = refs[b % 2]
        assert g["term"][b] == r.term == 0 and g["iters"][b] == r.iters and r.iters > 40
        assert np.array_equal(g["status"][b], r.status) and np.array_equal(g["x"][b], r.x) and np.array_equal(g["u"][b], r.u) and g["f"][b] == r.f
    g1 = gpu_solve(capi, [base[0]], max_iter=600)  # latency layout (batch of one)
    assert np.array_equal(g1["x"][0], refs[0].x) and g1["iters"][0] == refs[0].iters


def test_mpc_arrow_c5_full_size(oracle, SP, capi):
    """BASELINE config c5: 100 stages x 30 (+ head), 10 constraints per stage: n = 3030, m = 1000, ~590 iterations."""
    qp = SP.mpc_arrow_qp(nstage=100, nx=30, mc=10, seed=0, scale_a=1.0)
    g = check_batch(oracle, SP, capi, [qp], max_iter=1000, min_iters=300)
    assert g["term"][0] == 0


def test_fuzz_shapes_bit_exact(oracle, SP, capi):
    """Random structure / block partition / constraint partition / bounds mixes, small batches each: every output bit for bit."""
    rng = np.random.default_rng(2026)
    tot_iters = tot_drops = 0
    for case in range(24):
        kind = ["tri", "arrow_up", "arrow_down"][case % 3]
        nblk = int(rng.integers(2, 7))
        bs = [int(v) for v in rng.integers(1, 12, nblk)]
        n = sum(bs)
        if case % 2:  # C follows its own partition of the unknowns
            cuts = np.sort(rng.choice(np.arange(1, n), size=min(int(rng.integers(1, 4)), n - 1), replace=False)) if n > 1 else np.array([], int)
            cbs = [int(v) for v in np.diff(np.concatenate([[0], cuts, [n]]))]
        else:
            cbs = list(bs)
        mc = [int(rng.integers(0, 2 * c + 2)) for c in cbs]
        if sum(mc) == 0:
            mc[0] = 2
        box = [None, 1.0, 0.8][case % 3] if case % 4 else None
        qps = [SP.random_structured_qp(kind, bs, mc, 5000 + 10 * case + k, ridge=0.05, cbs=cbs, box=box, double_sided=bool(case % 5)) for k in range(4)]
        g = check_batch(oracle, SP, capi, qps)
        tot_iters += int(g["iters"].sum())
    assert tot_iters > 200


@pytest.mark.parametrize("kind,bs,mc", [("tri", [70, 128, 65, 96], [20, 30, 10, 25]), ("arrow_up", [96, 70, 128, 33], [15, 25, 30, 5]),
                                        ("arrow_down", [128, 40, 97, 65], [30, 10, 25, 12])])
def test_blocks_of_up_to_128_rows_bit_exact(oracle, SP, capi, kind, bs, mc):
    """Round 4: Hessian blocks of 65 .. 128 rows (two rows per lane in the block solves of L^-1 and L^-T, the factorisation through the
    panelled kernels) -- status, iterations, active set, x, u, f bit for bit; one batch also with C on its own partition and a box."""
    qps = [SP.random_structured_qp(kind, bs, mc, 900 + seed, ridge=0.05) for seed in range(4)]
    check_batch(oracle, SP, capi, qps, min_iters=40)
    n = sum(bs)
    qps = [SP.random_structured_qp(kind, bs, [40, 35, 20], 950 + seed, ridge=0.05, cbs=[100, n - 230, 130], box=1.0) for seed in range(3)]
    check_batch(oracle, SP, capi, qps, min_iters=20)


def test_beyond_3072_unknowns(oracle, SP, capi):
    """Round 4: n <= 4096 (a fourth chunk of 1024 chains per thread).  An arrow of 130 stages x 30 + head: n = 3930, m = 650."""
    qp = SP.mpc_arrow_qp(nstage=130, nx=30, mc=5, seed=3, scale_a=1.0)
    assert sum(qp.bs) == 3930
    g = check_batch(oracle, SP, capi, [qp], max_iter=1000, min_iters=100)
    assert g["term"][0] == 0


@pytest.mark.parametrize("form", ["sequence", "c9"])
@pytest.mark.parametrize("kind", ["tri", "arrow_up", "arrow_down"])
def test_both_forms_bit_exact(oracle, SP, capi, kind, form):
    """Round 5: each form on each shape, whatever AUTO would choose -- the reference's small problems, wider ones with removals, C blocks
    on their own partition with box bounds, vertex solutions (q == n) and a medium arrow / tridiagonal chain."""
    qps = [SP.random_structured_qp(kind, [3, 5, 2, 3], [3, 3, 3, 3], seed) for seed in range(12)]
    check_batch(oracle, SP, capi, qps, min_iters=12, form=form)
    assert FORM["form"] == {"sequence": 0, "c9": 1}[form]
    bs, mc = {"tri": ([7, 4, 9, 6, 5, 8], [4, 6, 3, 5, 7, 2]), "arrow_up": ([6, 9, 4, 7, 5], [5, 3, 6, 4, 2]), "arrow_down": ([8, 5, 7, 6, 9], [3, 6, 4, 5, 0])}[kind]
    check_batch(oracle, SP, capi, [SP.random_structured_qp(kind, bs, mc, 100 + seed, ridge=0.05) for seed in range(8)], min_iters=50, form=form)
    qps = [SP.random_structured_qp(kind, [6, 9, 4, 7, 5], [5, 7, 6], 300 + seed, ridge=0.05, cbs=[10, 8, 13], box=1.0) for seed in range(8)]
    check_batch(oracle, SP, capi, qps, min_iters=40, form=form)
    qps = []
    for seed in range(24):
        qp = SP.random_structured_qp(kind, [2, 1, 2], [4, 3, 4], 900 + seed, ridge=0.05, box=0.3)
        qp.a *= 30
        qps.append(qp)
    check_batch(oracle, SP, capi, qps, form=form)
    qps = [SP.random_structured_qp(kind, [30] * 12, [6] * 12, 40 + s, ridge=0.05, box=2.0) for s in range(3)]
    check_batch(oracle, SP, capi, qps, max_iter=800, min_iters=60, form=form)


@pytest.mark.parametrize("form", ["sequence", "c9"])
def test_c5_full_size_both_forms(oracle, SP, capi, form):
    qp = SP.mpc_arrow_qp(nstage=100, nx=30, mc=10, seed=1, scale_a=1.0)
    g = check_batch(oracle, SP, capi, [qp], max_iter=1000, min_iters=300, form=form)
    assert g["term"][0] == 0


def test_c9_blocks_of_up_to_128_rows_and_tridiagonal(oracle, SP, capi):
    """Contract C9 on shapes AUTO gives to the sequence form (tridiagonal) and on blocks of 65 .. 128 rows."""
    for kind, bs, mc in [("tri", [70, 128, 65, 96], [20, 30, 10, 25]), ("arrow_up", [96, 70, 128, 33], [15, 25, 30, 5]), ("arrow_down", [128, 40, 97, 65], [30, 10, 25, 12])]:
        qps = [SP.random_structured_qp(kind, bs, mc, 900 + seed, ridge=0.05) for seed in range(3)]
        check_batch(oracle, SP, capi, qps, min_iters=30, form="c9")
        n = sum(bs)
        qps = [SP.random_structured_qp(kind, bs, [40, 35, 20], 950 + seed, ridge=0.05, cbs=[100, n - 230, 130], box=1.0) for seed in range(2)]
        check_batch(oracle, SP, capi, qps, min_iters=10, form="c9")


@pytest.mark.parametrize("form", ["sequence", "c9"])
@pytest.mark.parametrize("kind", ["tri", "arrow_up", "arrow_down"])
def test_packed_block_storage_bit_exact(oracle, SP, capi, kind, form):
    """The blocks handed over back to back with their own leading dimensions (BlockGISolver.pack) instead of as views into N x N
    matrices: the same bits (the reference's StructuredG takes any MatrixRef per block)."""
    bs, mc = {"tri": ([7, 4, 9, 6, 5, 8], [4, 6, 3, 5, 7, 2]), "arrow_up": ([6, 9, 4, 7, 5], [5, 3, 6, 4, 2]), "arrow_down": ([8, 5, 7, 6, 9], [3, 6, 4, 5, 0])}[kind]
    check_batch(oracle, SP, capi, [SP.random_structured_qp(kind, bs, mc, 100 + seed, ridge=0.05) for seed in range(6)], min_iters=30, form=form, packed=True)
    qps = [SP.random_structured_qp(kind, [30] * 12, [6] * 12, 40 + s, ridge=0.05, box=2.0) for s in range(2)]
    check_batch(oracle, SP, capi, qps, max_iter=800, min_iters=40, form=form, packed=True)
