"""Planted-solution random QP generator (synthetic workload for tests and bench.py).

A RESTATEMENT, step for step, of the reference's test-support generator in numpy / scipy: the same construction in the same order (the
step numbering, the four branches and the variable names nVar / nstrong / rankTot / colsTot / red follow randomProblems.cpp so that the
two can be read side by side -- SURVEY.md 8(d) requires the recipe to be behaviourally identical; it is the workload generator, not the
hot path), with numpy / scipy primitives in place of Eigen's and a seeded counter-based RNG of its own (the reference draws from an
unseeded std::mt19937 / libc rand, so any fixed seed is admissible -- SURVEY.md §4, §8d):

  reference src/test/randomProblems.cpp:15-251   randomProblem(ProblemCharacteristics)
  reference include/jrl-qp/test/randomMatrices.h:62-215   randOrtho / randn / randDependent
  reference include/jrl-qp/test/problems.h:110-115, src/test/problems.cpp:7-27   least-squares -> QP

A problem  min 1/2 ||A x - b||^2  s.t.  E x = f,  l <= C x <= u,  xl <= x <= xu  is built around a
planted primal/dual solution (x, lambdaEq, lambdaIneq, lambdaBnd) satisfying the KKT conditions, then
handed to the solver as  G = A^T A,  a = -A^T b,  constraints stacked [E; C] and passed column-wise.

Seeding contract: problem `index` uses Philox key  SEED0 + index  (SEED0 = 0x6A726C71).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.linalg as sla

SEED0 = 0x6A726C71


def make_rng(index: int, seed0: int = SEED0) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=seed0 + int(index)))


def _uniform(rng, *shape):
    """Eigen's Random(): uniform in [-1, 1]."""
    return rng.uniform(-1.0, 1.0, size=shape)


def rand_ortho(rng, size: int) -> np.ndarray:
    """Haar-distributed orthogonal matrix (randMatrices.h:62-127 builds it from Householder
    reflectors; QR of a Gaussian matrix with the sign fix is the same distribution)."""
    if size == 0:
        return np.zeros((0, 0))
    q, r = np.linalg.qr(rng.standard_normal((size, size)))
    return q * np.sign(np.diag(r))


def randn(rng, rows: int, cols: int, rank: int = -1) -> np.ndarray:
    """Random matrix of a given rank with ~N(0,1) entries (randomMatrices.h:147-181)."""
    p = min(rows, cols)
    if rank < 0:
        rank = p
    if rank == 0:
        return np.zeros((rows, cols))
    if rank == p:
        return rng.standard_normal((rows, cols))
    s = np.zeros(p)
    s[:rank] = _uniform(rng, rank) * np.sqrt(3.0 * rows * cols / rank)
    m = np.zeros((rows, cols))
    if rows <= cols:
        m[:, :rows] = rand_ortho(rng, rows) * s
        return m @ rand_ortho(rng, cols)
    m[:cols, :] = s[:, None] * rand_ortho(rng, cols)
    return rand_ortho(rng, rows) @ m


def rand_dependent(rng, cols, rows_a, rank_a, rows_b, rank_b, rank_ab):
    """A (rows_a x cols, rank_a), B (rows_b x cols, rank_b) with rank([A;B]) = rank_ab
    (randomMatrices.h:189-215)."""
    m = randn(rng, rank_a + rank_b, cols, rank_ab)
    a = m[:rank_a] if rank_a == rows_a else rand_ortho(rng, rows_a)[:, :rank_a] @ m[:rank_a]
    b = m[rank_a:] if rank_b == rows_b else rand_ortho(rng, rows_b)[:, :rank_b] @ m[rank_a:]
    return a.copy(), b.copy()


@dataclass
class Characteristics:
    """Mirror of ProblemCharacteristics (include/jrl-qp/test/randomProblems.h:17-120)."""

    nVar: int
    nObj: int
    nEq: int = 0
    nIneq: int = 0
    rankObj: int = -1
    nSharedRank: int = 0
    nStrongActIneq: int = 0
    nWeakActIneq: int = 0
    nStrongActBounds: int = 0
    nWeakActBounds: int = 0
    bounds: bool = False
    doubleSidedIneq: bool = False
    strictlyFeasible: bool = False

    def __post_init__(self):
        if self.rankObj < 0:
            self.rankObj = self.nObj
        assert self.nVar >= self.nObj and self.nVar >= self.nEq
        assert self.nStrongActIneq + self.nWeakActIneq <= self.nIneq
        assert (self.bounds and self.nStrongActBounds + self.nWeakActBounds <= self.nVar) or (
            not self.bounds and self.nStrongActBounds == 0 and self.nWeakActBounds == 0
        )
        assert self.nEq + self.nStrongActIneq + self.nStrongActBounds <= self.nVar
        assert self.rankObj <= self.nObj and self.nSharedRank <= self.rankObj


@dataclass
class RandomLeastSquare:
    A: np.ndarray
    b: np.ndarray
    E: np.ndarray
    f: np.ndarray
    C: np.ndarray  # nIneq x nVar, one constraint per ROW (as in the reference generator)
    l: np.ndarray
    u: np.ndarray
    xl: np.ndarray
    xu: np.ndarray
    x: np.ndarray
    lambdaEq: np.ndarray
    lambdaIneq: np.ndarray
    lambdaBnd: np.ndarray
    bounds: bool = False
    doubleSidedIneq: bool = False

    def to_qp(self):
        """(G, a, C, bl, bu, xl, xu) as GoldfarbIdnaniSolver::solve takes them: C is n x m with one
        constraint per column, equalities first (problems.cpp:7-27, problems.h:110-115).
        C is returned Fortran-ordered so that .ravel(order='F') is the col-major buffer."""
        G = self.A.T @ self.A
        a = -self.A.T @ self.b
        Call = np.vstack([self.E, self.C])  # m x n
        bl = np.concatenate([self.f, self.l])
        bu = np.concatenate([self.f, self.u])
        return (
            np.asfortranarray(G),
            a.copy(),
            np.asfortranarray(Call.T),
            bl,
            bu,
            self.xl.copy(),
            self.xu.copy(),
        )

    def multipliers(self):
        return np.concatenate([self.lambdaEq, self.lambdaIneq, self.lambdaBnd])


def random_problem(ch: Characteristics, rng: np.random.Generator) -> RandomLeastSquare:
    """Restatement of randomProblem (randomProblems.cpp:15-251)."""
    nVar, nObj, nEq, nIneq = ch.nVar, ch.nObj, ch.nEq, ch.nIneq
    nSA, nWA, nSB, nWB = ch.nStrongActIneq, ch.nWeakActIneq, ch.nStrongActBounds, ch.nWeakActBounds
    nstrong = nEq + nSA + nSB
    nBounds = nVar if ch.bounds else 0
    colsTot = nObj + nstrong
    red = np.zeros(colsTot)

    # 1 - A, the strongly-active rows Ca and [u; v] with A^T u + Ca^T v = 0  (:24-84)
    if nObj == 0:
        A = np.zeros((0, nVar))
        Ca = randn(rng, nstrong, nVar)
    elif nstrong == 0:
        A = randn(rng, nObj, nVar, ch.rankObj)
        Ca = np.zeros((0, nVar))
    elif colsTot > nVar:
        rankTot = min(ch.rankObj + nstrong - ch.nSharedRank, nVar)
        A, Ca = rand_dependent(rng, nVar, nObj, ch.rankObj, nstrong, nstrong, rankTot)
        if nSB:
            Ca[nstrong - nSB :] = np.eye(nSB, nVar)
        M = np.hstack([A.T, Ca.T])  # nVar x colsTot
        _, R, P = sla.qr(M, pivoting=True)
        R1 = R[:rankTot, :rankTot]
        R2 = R[:rankTot, rankTot:]
        v = _uniform(rng, colsTot - rankTot)
        uu = sla.solve_triangular(R1, -R2 @ v, lower=False)
        red[P] = np.concatenate([uu, v])
    else:
        rankTot = ch.rankObj + nstrong - ch.nSharedRank
        if rankTot == nVar:
            rankTot = nVar - 1
        A, Ca = rand_dependent(rng, nVar, nObj, ch.rankObj, nstrong, nstrong, rankTot)
        if nSB:
            Ca[nstrong - nSB :] = np.eye(nSB, nVar)
        M = np.vstack([A, Ca])  # colsTot x nVar
        Q, _, _ = sla.qr(M, pivoting=True)
        # left null space of M (the reference takes Q.rightCols(nVar - rankTot), which coincides when colsTot == nVar)
        N = Q[:, rankTot:]
        red = N @ _uniform(rng, N.shape[1])

    # 2 - sign of the multipliers of one-sided inequalities (:87-101)
    if not ch.doubleSidedIneq:
        for i in range(nSA):
            if red[nObj + nEq + i] < 0:
                red[nObj + nEq + i] = -red[nObj + nEq + i]
                Ca[nEq + i] = -Ca[nEq + i]

    # 3 - containers (:104-125)
    E = Ca[:nEq].copy()
    lamEq = red[nObj : nObj + nEq].copy()
    C = np.zeros((nIneq, nVar))
    l = np.full(nIneq, -np.inf) if not ch.doubleSidedIneq else np.zeros(nIneq)
    lamIneq = np.zeros(nIneq)
    C[:nSA] = Ca[nEq : nEq + nSA]
    lamIneq[:nSA] = red[nObj + nEq : nObj + nEq + nSA]
    lamBnd = np.zeros(nBounds)
    if nSB:
        lamBnd[:nSB] = red[colsTot - nSB :]

    # 4 - weakly active and inactive rows (:128-156)
    if nWA:
        if nWA <= nstrong:
            Q1 = rand_ortho(rng, nstrong)[:nWA]
        else:
            Q1 = rand_ortho(rng, nWA)[:, :nstrong]
        if ch.strictlyFeasible:
            mult = red[colsTot - nstrong :]
            C[nSA : nSA + nWA] = (np.abs(Q1) * np.sign(mult)) @ Ca
        else:
            C[nSA : nSA + nWA] = Q1 @ Ca
    ninact = nIneq - nSA - nWA
    C[nSA + nWA :] = randn(rng, ninact, nVar)

    # 5 - solution and right-hand sides (:159-224)
    x = _uniform(rng, nVar)
    b = A @ x - red[:nObj]
    f = E @ x
    u = C @ x
    if ch.doubleSidedIneq:
        l = C @ x
        rl = np.abs(_uniform(rng, nIneq))
        ru = np.abs(_uniform(rng, nIneq))
        for i in range(nSA):
            if lamIneq[i] > 0:
                l[i] -= rl[i]
            else:
                u[i] += ru[i]
        for i in range(nSA, nSA + nWA):
            if rl[i] > ru[i]:
                l[i] -= rl[i]
            else:
                C[i] = -C[i]
                u[i] = -u[i] + ru[i]
                # NB the reference leaves l[i] = (C x)_i of the un-negated row here (:185-189); keep a valid pair instead
                l[i] = -l[i]
        if ninact:
            l[nIneq - ninact :] -= rl[nIneq - ninact :]
            u[nIneq - ninact :] += ru[nIneq - ninact :]
    else:
        if ninact:
            u[nIneq - ninact :] += np.abs(_uniform(rng, ninact))
    xl = np.zeros(nBounds)
    xu = np.zeros(nBounds)
    if ch.bounds:
        r = _uniform(rng, nVar)
        xl = x.copy()
        xu = x.copy()
        for i in range(nSB):
            if lamBnd[i] > 0:
                xl[i] -= abs(r[i])
            else:
                xu[i] += abs(r[i])
        for i in range(nSB, nSB + nWB):
            if r[i] > 0:
                xl[i] -= r[i]
            else:
                xu[i] -= r[i]
        inact = nVar - nSB - nWB
        if inact:
            xl[nVar - inact :] -= np.abs(_uniform(rng, inact))
            xu[nVar - inact :] += np.abs(_uniform(rng, inact))

    # 6 - Fisher-Yates shuffles (:226-249)
    for i in range(nIneq - 1, 0, -1):
        j = int(rng.integers(0, i, endpoint=True))
        C[[i, j]] = C[[j, i]]
        u[i], u[j] = u[j], u[i]
        lamIneq[i], lamIneq[j] = lamIneq[j], lamIneq[i]
        if ch.doubleSidedIneq:
            l[i], l[j] = l[j], l[i]
    if ch.bounds:
        for i in range(nVar - 1, 0, -1):
            j = int(rng.integers(0, i, endpoint=True))
            for Mx in (A, C, E):
                Mx[:, [i, j]] = Mx[:, [j, i]]
            for vx in (xl, xu, lamBnd, x):
                vx[i], vx[j] = vx[j], vx[i]

    return RandomLeastSquare(
        A=A, b=b, E=E, f=f, C=C, l=l, u=u, xl=xl, xu=xu, x=x, lambdaEq=lamEq, lambdaIneq=lamIneq, lambdaBnd=lamBnd,
        bounds=ch.bounds, doubleSidedIneq=ch.doubleSidedIneq,
    )


@dataclass
class PackedBatch:
    """B independent dense QPs packed contiguously, problem b at offset b*stride -- the layout the
    batched C-ABI (include/jrlqp_hip.h: jrlqp_gi_solve_batch) takes.  G[b] and C[b] are col-major:
    G[b].ravel() is the n*n col-major buffer, C[b] is stored as (m, n) C-order = n x m col-major."""

    n: int
    m: int
    nb: int
    G: np.ndarray  # (B, n, n)  symmetric, so row/col-major coincide
    a: np.ndarray  # (B, n)
    C: np.ndarray  # (B, m, n)  row i = normal of constraint i  == n x m col-major
    bl: np.ndarray  # (B, m)
    bu: np.ndarray  # (B, m)
    xl: np.ndarray  # (B, nb)
    xu: np.ndarray  # (B, nb)
    x_star: np.ndarray  # (B, n)   planted solution
    lam_star: np.ndarray  # (B, m+nb) planted multipliers
    meta: dict = field(default_factory=dict)

    @property
    def B(self):
        return self.G.shape[0]


def benchmark_characteristics(n: int, m: int, ma: int | None = None) -> Characteristics:
    """The reference benchmark recipe (benchmarks/Solvers.cpp:257-263): nObj = n, no equalities,
    m single-sided inequalities of which `ma` strongly active, no bounds.  Default ma = floor(0.3 n)
    (the 30 % active family, Solvers.cpp:621-623)."""
    if ma is None:
        ma = int(0.3 * n)
    return Characteristics(nVar=n, nObj=n, nEq=0, nIneq=m, nStrongActIneq=ma)


def generate_batch(B: int, ch: Characteristics, first_index: int = 0, seed0: int = SEED0) -> PackedBatch:
    n = ch.nVar
    m = ch.nEq + ch.nIneq
    nb = n if ch.bounds else 0
    G = np.empty((B, n, n))
    a = np.empty((B, n))
    C = np.empty((B, m, n))
    bl = np.empty((B, m))
    bu = np.empty((B, m))
    xl = np.empty((B, nb))
    xu = np.empty((B, nb))
    xs = np.empty((B, n))
    ls = np.empty((B, m + nb))
    for k in range(B):
        pb = random_problem(ch, make_rng(first_index + k, seed0))
        g, av, c, l, u, xlv, xuv = pb.to_qp()
        G[k] = g
        a[k] = av
        C[k] = c.T
        bl[k] = l
        bu[k] = u
        xl[k] = xlv
        xu[k] = xuv
        xs[k] = pb.x
        ls[k] = pb.multipliers()
    return PackedBatch(n=n, m=m, nb=nb, G=G, a=a, C=C, bl=bl, bu=bu, xl=xl, xu=xu, x_star=xs, lam_star=ls,
                       meta={"first_index": first_index, "seed0": seed0})


def box_and_single_constraint_problem(n: int, act: bool, rng, act_level: float = 0.5):
    """generateBoxAndSingleConstraintProblem (src/experimental/BoxAndSingleConstraintSolver.cpp:99-176):
    min 1/2 ||x - x0||^2  s.t.  c.x >= b, xl <= x <= xu, with x0, the box corners and c uniform in [-1, 1]^n
    (Eigen's VectorXd::Random).  act: b lies between c.xb (xb = x0 clamped to the box: where the constraint starts to be
    active) and c.su (su = the box vertex furthest along c: the last feasible value), at fraction act_level; otherwise
    b = c.sl (sl = the vertex closest along c), so that the whole box satisfies the constraint.
    Returns (x0, c, b, xl, xu)."""
    assert (not act) or (0.0 < act_level < 1.0)
    x0 = rng.uniform(-1.0, 1.0, n)
    r1 = rng.uniform(-1.0, 1.0, n)
    r2 = rng.uniform(-1.0, 1.0, n)
    xl, xu = np.minimum(r1, r2), np.maximum(r1, r2)
    xb = np.clip(x0, xl, xu)
    c = rng.uniform(-1.0, 1.0, n)
    sl = np.where(c > 0, xl, xu)
    su = np.where(c > 0, xu, xl)
    if act:
        d1, d2 = float(c @ xb), float(c @ su)
        b = act_level * d1 + (1.0 - act_level) * d2
    else:
        b = float(c @ sl)
    return x0, c, b, xl, xu
