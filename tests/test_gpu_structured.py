"""GPU parity tests of the structured Cholesky family (jrlqp_blocks_* of the C ABI) against the CPU oracle --
bit for bit -- on the reference's own test shapes ({3,5,2,3}, every zero window; tests/blockArrowLLTTest.cpp,
tests/triBlockDiagLLTTest.cpp), on the MultiIK fixture matrices and on the BASELINE c5 shape (100 x 30)."""
import numpy as np
import pytest

from ikdata import read_arrow, read_tri_block_diag

pytestmark = pytest.mark.gpu

BS = [3, 5, 2, 3]
OFF = np.concatenate([[0], np.cumsum(BS)])
N = int(OFF[-1])


@pytest.fixture(scope="module")
def capi(pkg):
    import jrl_qp_amd.capi as capi

    return capi


def views(H, bs, kind):
    off = np.concatenate([[0], np.cumsum(bs)]).astype(int)
    nb = len(bs)
    D = [H[off[i] : off[i + 1], off[i] : off[i + 1]] for i in range(nb)]
    if kind == "arrow_down":
        S = [H[off[nb - 1] : off[nb], off[i] : off[i + 1]] for i in range(nb - 1)]
    elif kind == "arrow_up":
        S = [H[off[i + 1] : off[i + 2], 0 : bs[0]] for i in range(nb - 1)]
    else:
        S = [H[off[i + 1] : off[i + 2], off[i] : off[i + 1]] for i in range(nb - 1)]
    return D, S


def spd(rng, bs, kind):
    off = np.concatenate([[0], np.cumsum(bs)]).astype(int)
    n = int(off[-1])
    A = np.zeros((n, n))
    nb = len(bs)
    for i in range(nb):
        k = off[i]
        A[k : k + bs[i], k : k + bs[i]] = rng.uniform(-1, 1, (bs[i], bs[i]))
        if kind == "arrow_down":
            A[k : k + bs[i], n - bs[-1] :] = rng.uniform(-1, 1, (bs[i], bs[-1]))
        elif kind == "arrow_up":
            A[k : k + bs[i], : bs[0]] = rng.uniform(-1, 1, (bs[i], bs[0]))
        elif i + 1 < nb:
            A[off[i + 1] : off[i + 2], k : k + bs[i]] = rng.uniform(-1, 1, (bs[i + 1], bs[i]))
    H = A.T @ A if kind != "tridiag" else A @ A.T
    return np.asfortranarray(H + 1e-3 * np.eye(n))


def to_dev(Hs):
    """list of column-major (n, n) numpy matrices -> (nprob, n, n) CUDA tensor whose [b].T is the matrix"""
    import torch

    return torch.from_numpy(np.ascontiguousarray(np.stack([h.T for h in Hs]))).cuda()


def from_dev(T):
    return [np.asfortranarray(t.T) for t in T.cpu().numpy()]


def oracle_factor(oracle, H, bs, kind):
    Hc = np.asfortranarray(H.copy())
    D, S = views(Hc, bs, kind)
    ok = oracle.block_arrow_llt(D, S, up=(kind == "arrow_up")) if kind != "tridiag" else oracle.tri_block_diag_llt(D, S)
    return ok, Hc


def rhs_dev(Ms):
    import torch

    return torch.from_numpy(np.ascontiguousarray(np.stack([m.T for m in Ms]))).cuda()


@pytest.mark.parametrize("kind", ["arrow_down", "arrow_up", "tridiag"])
def test_reference_shapes_all_windows(capi, oracle, kind):
    import torch

    rng = np.random.default_rng({"arrow_down": 1, "arrow_up": 2, "tridiag": 3}[kind])
    H = spd(rng, BS, kind)
    ok_ref, Href = oracle_factor(oracle, H, BS, kind)
    T = to_dev([H])
    plan = capi.StructuredBlocks(kind, BS, T, max_nrhs=5)
    ok = plan.llt()
    torch.cuda.synchronize()
    assert ok == [ok_ref] and ok_ref
    Hg = from_dev(T)[0]
    assert np.array_equal(Hg, Href)  # factor in place, bit-identical, untouched entries included
    Dr, Sr = views(Href, BS, kind)
    for i in range(N):
        for j in range(i + 1, N + 1):
            B = np.zeros((N, 5), order="F")
            B[i:j] = rng.uniform(-1, 1, (j - i, 5))
            for transpose in (False, True):
                for windowed in (False, True):
                    st, en = (i, j) if windowed else (0, -1)
                    ref = B.copy(order="F")
                    if kind == "tridiag":
                        oracle.tri_block_diag_solve(Dr, Sr, ref, start=st, end=en, transpose=transpose)
                    else:
                        oracle.block_arrow_solve(Dr, Sr, kind == "arrow_up", ref, st, en, transpose=transpose)
                    M = rhs_dev([B])
                    plan.solve(M, transpose=transpose, start=st, end=en)
                    got = M.cpu().numpy()[0].T
                    assert np.array_equal(got, ref), (kind, i, j, transpose, windowed)


@pytest.mark.parametrize("kind", ["arrow_down", "tridiag"])
def test_c5_shape_batch(capi, oracle, kind):
    """BASELINE config c5 shape: 100 blocks of 30 (+ a 30-row head for the arrow), a batch of 4 problems."""
    import torch

    rng = np.random.default_rng(55)
    bs = [30] * (101 if kind == "arrow_down" else 100)
    Hs = [spd(rng, bs, kind) for _ in range(4)]
    refs = [oracle_factor(oracle, h, bs, kind) for h in Hs]
    T = to_dev(Hs)
    plan = capi.StructuredBlocks(kind, bs, T, max_nrhs=2)
    ok = plan.llt()
    assert ok == [r[0] for r in refs] and all(ok)
    for hg, (_, hr) in zip(from_dev(T), refs):
        assert np.array_equal(hg, hr)
    n = sum(bs)
    Bs = [np.asfortranarray(rng.standard_normal((n, 2))) for _ in range(4)]
    M = rhs_dev(Bs)
    plan.solve(M, transpose=False)
    plan.solve(M, transpose=True)
    torch.cuda.synchronize()
    got = M.cpu().numpy()
    for b in range(4):
        ref = Bs[b].copy(order="F")
        D, S = views(refs[b][1], bs, kind)
        if kind == "tridiag":
            oracle.tri_block_diag_solve(D, S, ref)
            oracle.tri_block_diag_solve(D, S, ref, transpose=True)
        else:
            oracle.block_arrow_solve(D, S, False, ref)
            oracle.block_arrow_solve(D, S, False, ref, transpose=True)
        assert np.array_equal(got[b].T, ref)
        assert np.allclose(Hs[b] @ got[b].T, Bs[b], atol=1e-7)  # and it does solve H x = b


def test_multiik_fixture_matrices(capi, oracle):
    """The Hessians of the reference's MultiIK fixtures: block-tridiagonal 9 x 43 and block-arrow(up) 5 x 42
    (tests/BlockGISolverTest.in.cpp:196-214, :288-300)."""
    d = read_tri_block_diag()
    for kind, H, bs in (("tridiag", d["G"], [43] * 9), ("arrow_up", read_arrow()["G"], [42] * 5)):
        H = np.asfortranarray(H)
        ok_ref, Href = oracle_factor(oracle, H, bs, kind)
        T = to_dev([H])
        plan = capi.StructuredBlocks(kind, bs, T, max_nrhs=1)
        assert plan.llt() == [ok_ref] and ok_ref
        assert np.array_equal(from_dev(T)[0], Href)


def test_failure_flags(capi, oracle):
    """Not positive definite: detected at a column > 0 of a block (the reference's `ret > 0` test)."""
    H = np.asfortranarray(np.diag([1.0, -2.0, 3.0, 4.0]))
    for kind in ("tridiag", "arrow_down"):
        ok_ref, _ = oracle_factor(oracle, H, [2, 2], kind)
        T = to_dev([H, np.asfortranarray(np.eye(4))])
        plan = capi.StructuredBlocks(kind, [2, 2], T, max_nrhs=1)
        assert plan.llt() == [ok_ref, True] and not ok_ref


def random_structured_cases(capi, o, ncases, seed, verbose=False, tops=(8, 32, 48, 64), max_blocks=8):
    """Random block counts and sizes (1..64: register kernel up to 48 rows, barrier kernel beyond), batches, indefinite blocks at
    random positions (column 0 of a block -- the failure the reference does not notice -- and later columns), and, after a
    successful factorisation, L and L^T solves of random right-hand sides with random zero windows; returns the number of
    cases that differ from the oracle in any bit."""
    import torch

    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(ncases):
        kind = ["tridiag", "tridiag", "arrow_down", "arrow_up"][case % 4]
        nblk = int(rng.integers(1, max_blocks + 1))
        top = int(rng.choice(list(tops)))
        bs = [int(rng.integers(1, top + 1)) for _ in range(nblk)]
        nprob = int(rng.choice([1, 2, 5]))
        Hs = []
        for b in range(nprob):
            H = spd(rng, bs, kind)
            if rng.random() < 0.35:  # make one diagonal entry hopeless: a failure at column j of block i
                i = int(rng.integers(0, nblk)); j = int(rng.integers(0, bs[i])); k = int(sum(bs[:i]) + j)
                H[k, k] = -abs(H[k, k]) - 1.0
            Hs.append(H)
        refs = [oracle_factor(o, H, bs, kind) for H in Hs]
        T = to_dev(Hs)
        plan = capi.StructuredBlocks(kind, bs, T, max_nrhs=2)
        ok = plan.llt(); torch.cuda.synchronize()
        Hg = from_dev(T)
        # (a failing ARROW factorisation: the flag only -- the blocks are factorised in parallel here, one after the other in the
        # reference, so what a failure leaves behind in the blocks after it differs; the tridiagonal chain is compared in full)
        good = ok == [r[0] for r in refs] and all(np.array_equal(Hg[b], refs[b][1], equal_nan=True) for b in range(nprob) if ok[b] or kind == "tridiag")
        if good and all(ok):
            n = sum(bs); off = np.concatenate([[0], np.cumsum(bs)]).astype(int)
            i = int(rng.integers(0, nblk)); j = int(rng.integers(i + 1, nblk + 1))
            Bs = []
            for b in range(nprob):
                B = np.zeros((n, 2), order="F"); B[off[i]:off[j]] = rng.uniform(-1, 1, (off[j] - off[i], 2)); Bs.append(B)
            for transpose in (False, True):
                for windowed in (False, True):
                    st, en = (int(off[i]), int(off[j])) if windowed else (0, -1)  # rows of the non-zero window
                    M = rhs_dev(Bs)
                    plan.solve(M, transpose=transpose, start=st, end=en)
                    torch.cuda.synchronize()
                    Mg = M.cpu().numpy()
                    for b in range(nprob):
                        ref = Bs[b].copy(order="F"); Dr, Sr = views(refs[b][1], bs, kind)
                        if kind == "tridiag":
                            o.tri_block_diag_solve(Dr, Sr, ref, start=st, end=en, transpose=transpose)
                        else:
                            o.block_arrow_solve(Dr, Sr, kind == "arrow_up", ref, st, en, transpose=transpose)
                        good = good and np.array_equal(Mg[b].T, ref, equal_nan=True)
        if not good:
            bad += 1
            if verbose: print(f"MISMATCH case {case}: kind {kind} bs {bs} nprob {nprob} ok {ok} ref {[r[0] for r in refs]}", flush=True)
    return bad


def test_random_shapes_and_failures(capi, oracle):
    assert random_structured_cases(capi, oracle, 48, 2024) == 0


def test_random_shapes_and_failures_blocks_up_to_128_rows(capi, oracle):
    """Round 4: blocks of 65 .. 128 rows (two rows per lane in the solves, side blocks through LDS in panels, one block region
    in the tridiagonal chain) -- same generator, block sizes up to 96 / 128, failures included."""
    assert random_structured_cases(capi, oracle, 24, 4096, tops=(96, 128), max_blocks=5) == 0


@pytest.mark.parametrize("kind", ["arrow_down", "arrow_up", "tridiag"])
def test_big_blocks_unaligned_windows(capi, oracle, kind):
    """Block sizes 70 / 128 / 65 / 96: the factor bit for bit, then L and L^T solves with zero windows that start and end INSIDE
    blocks (on both sides of row 64 of a block), three right-hand sides."""
    import torch

    bs = [70, 128, 65, 96]
    off = np.concatenate([[0], np.cumsum(bs)]).astype(int)
    n = int(off[-1])
    rng = np.random.default_rng({"arrow_down": 11, "arrow_up": 12, "tridiag": 13}[kind])
    H = spd(rng, bs, kind)
    ok_ref, Href = oracle_factor(oracle, H, bs, kind)
    T = to_dev([H])
    plan = capi.StructuredBlocks(kind, bs, T, max_nrhs=3)
    ok = plan.llt()
    torch.cuda.synchronize()
    assert ok == [ok_ref] and ok_ref
    assert np.array_equal(from_dev(T)[0], Href)
    Dr, Sr = views(Href, bs, kind)
    wins = [(0, n), (3, 60), (10, 75), (66, 130), (71, 199), (140, 198), (135, 270), (200, 264), (263, 359), (300, 340), (330, n), (69, 71)]
    for st, en in wins:
        B = np.zeros((n, 3), order="F")
        B[st:en] = rng.uniform(-1, 1, (en - st, 3))
        for transpose in (False, True):
            for windowed in (False, True):
                a, b = (st, en) if windowed else (0, -1)
                ref = B.copy(order="F")
                if kind == "tridiag":
                    oracle.tri_block_diag_solve(Dr, Sr, ref, start=a, end=b, transpose=transpose)
                else:
                    oracle.block_arrow_solve(Dr, Sr, kind == "arrow_up", ref, a, b, transpose=transpose)
                M = rhs_dev([B])
                plan.solve(M, transpose=transpose, start=a, end=b)
                got = M.cpu().numpy()[0].T
                assert np.array_equal(got, ref), (kind, st, en, transpose, windowed)
