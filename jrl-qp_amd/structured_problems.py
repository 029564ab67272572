"""Synthetic problems for the structured GI path (block-tridiagonal / block-arrow Hessian, block-diagonal constraints).

Patterns follow the reference's own structured tests:
  tests/BlockGISolverTest.in.cpp:27-68   biBlockDiagRandom / blockDiagAndOneColDiagRandom (G = A A^T or A^T A)
  tests/BlockGISolverTest.in.cpp:70-170  per-block double-sided constraints taken from randomProblem(n_i, 0, 0, m_i)
  SURVEY.md §8(d) c5                      MPC-style block arrow: 100 stages x 30 (+ head), 10 constraints per stage
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import problems as P


@dataclass
class StructuredQP:
    kind: str  # "tri" | "arrow_up" | "arrow_down"
    bs: list  # block sizes of G
    G: np.ndarray  # dense n x n (Fortran order); the structured solvers factorise views of a COPY of it
    a: np.ndarray
    Cblocks: list  # block-diagonal C: list of n_i x m_i arrays (one constraint per column)
    bl: np.ndarray
    bu: np.ndarray
    xl: np.ndarray | None = None
    xu: np.ndarray | None = None
    meta: dict = field(default_factory=dict)

    @property
    def n(self):
        return int(sum(self.bs))

    @property
    def m(self):
        return int(sum(c.shape[1] for c in self.Cblocks))

    def dense_C(self):
        C = np.zeros((self.n, self.m), order="F")
        r = c = 0
        for blk in self.Cblocks:
            C[r : r + blk.shape[0], c : c + blk.shape[1]] = blk
            r += blk.shape[0]
            c += blk.shape[1]
        return C


def block_views(G, bs, kind):
    """(diag, off) lists of col-major VIEWS into G, as the reference's tests pack them (BlockGISolverTest.in.cpp:78-81,
    :130-133, :212-221, :293-302).  tri: off[i] = G[blk i+1, blk i]; arrow_up: off[i] = G[blk i+1, blk 0];
    arrow_down: off[i] = G[blk last, blk i]."""
    assert G.flags.f_contiguous
    o = np.concatenate([[0], np.cumsum(bs)]).astype(int)
    k = len(bs)
    diag = [G[o[i] : o[i + 1], o[i] : o[i + 1]] for i in range(k)]
    if kind == "tri":
        off = [G[o[i + 1] : o[i + 2], o[i] : o[i + 1]] for i in range(k - 1)]
    elif kind == "arrow_up":
        off = [G[o[i + 1] : o[i + 2], o[0] : o[1]] for i in range(k - 1)]
    elif kind == "arrow_down":
        off = [G[o[k - 1] : o[k], o[i] : o[i + 1]] for i in range(k - 1)]
    else:
        raise ValueError(kind)
    return diag, off


def _pattern(rng, bs, kind):
    """Random square factor A with the block pattern whose Gram matrix has the requested structure."""
    o = np.concatenate([[0], np.cumsum(bs)]).astype(int)
    n, k = int(o[-1]), len(bs)
    A = np.zeros((n, n))
    for i in range(k):
        A[o[i] : o[i + 1], o[i] : o[i + 1]] = rng.uniform(-1, 1, (bs[i], bs[i]))
        if kind == "tri" and i + 1 < k:
            A[o[i + 1] : o[i + 2], o[i] : o[i + 1]] = rng.uniform(-1, 1, (bs[i + 1], bs[i]))
        elif kind == "arrow_up":
            A[o[i] : o[i + 1], o[0] : o[1]] = rng.uniform(-1, 1, (bs[i], bs[0]))
        elif kind == "arrow_down":
            A[o[i] : o[i + 1], o[k - 1] : o[k]] = rng.uniform(-1, 1, (bs[i], bs[-1]))
    return A


def random_structured_qp(kind, bs, mc, seed, double_sided=True, ridge=0.0, cbs=None, box=None) -> StructuredQP:
    """G with the requested block structure (tri: A A^T, arrows: A^T A, as the reference's tests), a ~ U[-1,1],
    per-block constraints C_i (n_i x mc[i]) from the planted-solution generator with no objective
    (ProblemCharacteristics(n_i, 0, 0, m_i).doubleSidedIneq(true), BlockGISolverTest.in.cpp:93-101).
    cbs: row partition of the block-diagonal C when it is NOT the partition of G (StructuredC only needs the totals to
    match: constraint windows then straddle blocks of G); box: half-width of box bounds xl = -box, xu = +box, or None."""
    rng = np.random.default_rng(seed)
    A = _pattern(rng, bs, kind)
    G = A @ A.T if kind == "tri" else A.T @ A
    n = G.shape[0]
    if ridge:
        G = G + ridge * np.eye(n)
    G = np.asfortranarray(0.5 * (G + G.T))
    a = rng.uniform(-1, 1, n)
    Cb, bl, bu = [], [], []
    assert cbs is None or sum(cbs) == sum(bs)
    for ni, mi in zip(bs if cbs is None else cbs, mc):
        if mi == 0:
            Cb.append(np.zeros((ni, 0), order="F"))
            continue
        ch = P.Characteristics(nVar=ni, nObj=0, nEq=0, nIneq=mi, doubleSidedIneq=double_sided)
        pb = P.random_problem(ch, rng)
        Cb.append(np.asfortranarray(pb.C.T))
        bl.append(pb.l)
        bu.append(pb.u)
    xl = xu = None
    if box is not None:
        xl, xu = np.full(n, -float(box)), np.full(n, float(box))
    cat = lambda v: np.concatenate(v) if v else np.zeros(0)
    return StructuredQP(kind, list(bs), G, a, Cb, cat(bl), cat(bu), xl, xu, meta=dict(seed=seed))


def mpc_arrow_qp(nstage=100, nx=30, mc=10, seed=0, scale_a=4.0) -> StructuredQP:
    """BASELINE config c5: block arrow (pointing down) with `nstage` stage blocks of size nx plus a head block of size nx
    (G = M^T M, M with the arrow pattern of tests/blockArrowLLTTest.cpp:17-37), `mc` one-sided constraints per stage
    block (block-diagonal C, pattern of BlockGISolverTest.in.cpp:85-101), none on the head.  `scale_a` pushes the
    unconstrained optimum outside the feasible set so that a fraction of the constraints becomes active."""
    rng = np.random.default_rng(0xC5 + seed)
    bs = [nx] * nstage + [nx]
    A = _pattern(rng, bs, "arrow_down")
    G = np.asfortranarray(A.T @ A)
    G = np.asfortranarray(0.5 * (G + G.T))
    n = G.shape[0]
    a = scale_a * rng.uniform(-1, 1, n)
    Cb, bl, bu = [], [], []
    for i in range(nstage):
        Ci = rng.standard_normal((nx, mc))
        x0 = rng.uniform(-1, 1, nx)
        ub = Ci.T @ x0 + np.abs(rng.uniform(0, 1, mc))  # x0 strictly feasible
        Cb.append(np.asfortranarray(Ci))
        bl.append(np.full(mc, -np.inf))
        bu.append(ub)
    Cb.append(np.zeros((nx, 0), order="F"))
    return StructuredQP("arrow_down", bs, G, a, Cb, np.concatenate(bl), np.concatenate(bu), meta=dict(seed=seed))


def mpc_arrow_packed(nstage=100, nx=30, mc=10, seed=0, scale_a=4.0):
    """The problem of mpc_arrow_qp(nstage, nx, mc, seed, scale_a) WITHOUT ever forming the n x n Hessian: the same random factor M (same
    generator, same draws), the blocks of G = M^T M formed block by block and returned in the packed layout of
    capi.BlockGISolver.pack (diagonal blocks, then the side blocks G[head, stage i], each column-major with its own leading dimension:
    1.43 MB instead of 73 MB at c5).  The blocks agree with mpc_arrow_qp's to rounding (a 30-term product summed by another BLAS call),
    so this is a problem of the same family, not bit for bit the same instance.  Returns (packed, a, Cflat, bl, bu, bs, cshapes)."""
    rng = np.random.default_rng(0xC5 + seed)
    k = nstage + 1
    # the draws of _pattern(rng, bs, "arrow_down"): per block i the diagonal block, then the block in the head's columns
    Aii, Aih = [], []
    for i in range(k):
        Aii.append(rng.uniform(-1, 1, (nx, nx)))
        Aih.append(rng.uniform(-1, 1, (nx, nx)))  # (for i = k - 1 this draw overwrites the diagonal block of the head: as in _pattern)
    Aii[k - 1] = Aih[k - 1]
    diag, side = [], []
    head = Aii[k - 1].T @ Aii[k - 1]
    for i in range(nstage):
        diag.append(Aii[i].T @ Aii[i])
        side.append(Aih[i].T @ Aii[i])  # G[head, stage i] = A[i, head]^T A[i, i]
        head = head + Aih[i].T @ Aih[i]
    diag.append(head)
    diag = [0.5 * (d + d.T) for d in diag]
    n = k * nx
    a = scale_a * rng.uniform(-1, 1, n)
    Cb, bu = [], []
    for i in range(nstage):
        Ci = rng.standard_normal((nx, mc))
        x0 = rng.uniform(-1, 1, nx)
        bu.append(Ci.T @ x0 + np.abs(rng.uniform(0, 1, mc)))
        Cb.append(Ci)
    packed = np.concatenate([d.ravel(order="F") for d in diag] + [s_.ravel(order="F") for s_ in side])
    Cflat = np.concatenate([c.ravel(order="F") for c in Cb])
    bu = np.concatenate(bu)
    return packed, a, Cflat, np.full(bu.shape, -np.inf), bu, [nx] * k, [(nx, mc)] * nstage + [(nx, 0)]
