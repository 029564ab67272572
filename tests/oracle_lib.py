"""ctypes binding of the CPU oracle (oracle/libjrlqp_oracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libjrlqp_oracle.so")

c_dp = ct.POINTER(ct.c_double)
c_ip = ct.POINTER(ct.c_int)


class TraceRec(ct.Structure):
    _fields_ = [("p", ct.c_int), ("status", ct.c_int), ("l", ct.c_int), ("kind", ct.c_int), ("t1", ct.c_double), ("t2", ct.c_double)]


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("jrlqp_oracle.c", "structured_oracle.c", "blockgi_oracle.c", "jrlqp_oracle.h", "Makefile")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libjrlqp_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_ip) if a is not None else None


@dataclass
class OracleResult:
    term: int
    x: np.ndarray
    u: np.ndarray
    status: np.ndarray
    active: np.ndarray
    iters: int
    f: float
    trace: list
    L: np.ndarray  # G after the solve (lower triangle = Cholesky factor)


class Oracle:
    def __init__(self):
        self.lib = ct.CDLL(build_oracle())
        L = self.lib
        L.orc_gi_work_doubles.restype = ct.c_long
        L.orc_gi_solve.restype = ct.c_int
        L.orc_gi_solve.argtypes = [ct.c_int, ct.c_int, ct.c_int, c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_dp, c_dp, c_dp, c_dp,
                                   ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip, c_dp, ct.POINTER(TraceRec), ct.c_int, c_dp]
        L.orc_gi_solve_batch.restype = None
        L.orc_gi_solve_batch.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                         ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_dp, ct.c_int]
        L.orc_gi_solve_batch_literal.restype = None
        L.orc_gi_solve_batch_literal.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp,
                                                 ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip, c_ip, c_dp, ct.c_int]
        L.orc_gi_bench_rounds.restype = ct.c_double
        L.orc_gi_bench_rounds.argtypes = [ct.c_int] * 5 + [c_dp] * 7 + [ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_dp, ct.c_int]
        L.orc_gi_solve_warm.restype = ct.c_int
        L.orc_gi_solve_warm.argtypes = [ct.c_int, ct.c_int, ct.c_int, c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_dp, c_dp, c_dp, c_dp, c_ip,
                                        ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip, c_dp, ct.POINTER(TraceRec), ct.c_int, c_dp]
        L.orc_gi_solve_batch_warm.restype = None
        L.orc_gi_solve_batch_warm.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_ip,
                                              ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_dp, ct.c_int]
        L.orc_box_solve.restype = ct.c_int
        L.orc_box_solve.argtypes = [ct.c_int, c_dp, c_dp, ct.c_double, c_dp, c_dp, ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip,
                                    c_dp, ct.POINTER(TraceRec), ct.c_int]
        L.orc_llt.restype = ct.c_int
        L.orc_llt.argtypes = [ct.c_int, c_dp, ct.c_int]
        L.orc_inv_lt.restype = None
        L.orc_inv_lt.argtypes = [ct.c_int, c_dp, ct.c_int, c_dp, ct.c_int]
        L.orc_givens.restype = None
        L.orc_givens.argtypes = [ct.c_double, ct.c_double, c_dp, c_dp, c_dp]
        L.orc_dot.restype = ct.c_double
        L.orc_dot.argtypes = [ct.c_int, c_dp, c_dp]
        L.orc_dot_blk.restype = ct.c_double
        L.orc_dot_blk.argtypes = [ct.c_int, c_dp, c_dp, ct.c_int]
        L.orc_as_new.restype = ct.c_void_p
        L.orc_as_new.argtypes = [ct.c_int, ct.c_int]
        for nm in ("orc_as_free", "orc_as_reset"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = [ct.c_void_p]
        L.orc_as_activate.restype = None
        L.orc_as_activate.argtypes = [ct.c_void_p, ct.c_int, ct.c_int]
        L.orc_as_deactivate.restype = None
        L.orc_as_deactivate.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_as_status.restype = ct.c_int
        L.orc_as_status.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_as_at.restype = ct.c_int
        L.orc_as_at.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_as_counters.restype = ct.c_int
        L.orc_as_counters.argtypes = [ct.c_void_p, c_ip]
        pp = ct.POINTER(c_dp)
        blk = [ct.c_int, c_ip, pp, c_ip, pp, c_ip]
        L.orc_block_arrow_llt.restype = ct.c_int
        L.orc_block_arrow_llt.argtypes = blk + [ct.c_int]
        for nm in ("orc_block_arrow_lsolve", "orc_block_arrow_ltsolve"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = blk + [ct.c_int, c_dp, ct.c_int, ct.c_int, ct.c_int, ct.c_int]
        L.orc_tri_block_diag_llt.restype = ct.c_int
        L.orc_tri_block_diag_llt.argtypes = blk
        for nm in ("orc_tri_block_diag_lsolve", "orc_tri_block_diag_ltsolve"):
            getattr(L, nm).restype = None
            getattr(L, nm).argtypes = blk + [c_dp, ct.c_int, ct.c_int, ct.c_int]
        L.orc_blockgi_solve.restype = ct.c_int
        L.orc_blockgi_solve.argtypes = [ct.c_int] + blk + [c_dp, ct.c_int, c_ip, c_ip, pp, c_ip, c_dp, c_dp, c_dp, c_dp, ct.c_int,
                                                           ct.c_int, ct.c_double, c_dp, c_dp, c_ip, c_ip, c_ip, c_ip, c_dp,
                                                           ct.POINTER(TraceRec), ct.c_int]
        L.orc_set_vector_trace.restype = None
        L.orc_set_vector_trace.argtypes = [ct.c_void_p, ct.c_int]
        L.orc_blockgi_solve_c9.restype = ct.c_int
        L.orc_blockgi_solve_c9.argtypes = L.orc_blockgi_solve.argtypes
        L.orc_dot_tree.restype = ct.c_double
        L.orc_dot_tree.argtypes = [ct.c_int, c_dp, c_dp, ct.c_int]
        L.orc_oseq_new.restype = ct.c_void_p
        L.orc_oseq_new.argtypes = [ct.c_int]
        L.orc_oseq_free.restype = None
        L.orc_oseq_free.argtypes = [ct.c_void_p]
        L.orc_oseq_add_householder.restype = ct.c_double
        L.orc_oseq_add_householder.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, c_dp]
        L.orc_oseq_add_givens.restype = None
        L.orc_oseq_add_givens.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, c_dp, c_dp]
        L.orc_oseq_apply.restype = None
        L.orc_oseq_apply.argtypes = [ct.c_void_p, c_dp, ct.c_int]

    # ---- dense GI ------------------------------------------------------------------------------
    def solve(self, G, a, C, bl, bu, xl=None, xu=None, max_iter=500, big_bnd=1e100, trace_cap=0, warm=False, guess=None) -> OracleResult:
        """G n x n, C n x m (one constraint per column); inputs are copied (G is clobbered by the solver).
        warm=True: experimental::GoldfarbIdnaniSolver with warmStart_, `guess` = activation-status vector or None."""
        Gf = np.array(G, dtype=np.float64, order="F", copy=True)
        n = Gf.shape[0]
        Cf = np.asfortranarray(np.asarray(C, dtype=np.float64).reshape(n, -1))
        m = Cf.shape[1]
        a = np.ascontiguousarray(a, dtype=np.float64)
        bl = np.ascontiguousarray(bl, dtype=np.float64)
        bu = np.ascontiguousarray(bu, dtype=np.float64)
        nb = 0 if xl is None else len(xl)
        xl = np.ascontiguousarray(xl, dtype=np.float64) if nb else None
        xu = np.ascontiguousarray(xu, dtype=np.float64) if nb else None
        x = np.zeros(n)
        u = np.zeros(m + nb)
        status = np.zeros(m + nb, dtype=np.int32)
        active = np.zeros(m + nb + 1, dtype=np.int32)
        nact = ct.c_int(0)
        iters = ct.c_int(0)
        f = ct.c_double(0.0)
        tr = (TraceRec * max(trace_cap, 1))()
        if warm:
            g = None if guess is None else np.ascontiguousarray(guess, dtype=np.int32)
            assert g is None or g.shape == (m + nb,)
            term = self.lib.orc_gi_solve_warm(n, m, nb, _dp(Gf), max(n, 1), _dp(a), _dp(Cf), max(n, 1), _dp(bl), _dp(bu), _dp(xl),
                                              _dp(xu), _ip(g), max_iter, big_bnd, _dp(x), _dp(u), _ip(status), _ip(active),
                                              ct.byref(nact), ct.byref(iters), ct.byref(f), tr, trace_cap, None)
        else:
            term = self.lib.orc_gi_solve(n, m, nb, _dp(Gf), max(n, 1), _dp(a), _dp(Cf), max(n, 1), _dp(bl), _dp(bu), _dp(xl), _dp(xu),
                                         max_iter, big_bnd, _dp(x), _dp(u), _ip(status), _ip(active), ct.byref(nact), ct.byref(iters),
                                         ct.byref(f), tr, trace_cap, None)
        trace = [(t.p, t.status, t.l, t.kind, t.t1, t.t2) for t in tr[: min(trace_cap, iters.value)]]
        return OracleResult(term, x, u, status, active[: nact.value].copy(), iters.value, f.value, trace, Gf)

    def solve_with_vectors(self, G, a, C, bl, bu, xl=None, xu=None, cap=64, **kw):
        """solve() with the per-pass vector trace on: returns (result, [dict(x, u, z, r, f, q)] per pass) -- the values the reference
        logs under ITERATION_BASIC_DETAILS (DualSolver.cpp:108, :128-129), in the layout of jrlqp_dense_set_trace_vectors."""
        C = np.asarray(C)
        n, m = np.asarray(G).shape[0], (C.shape[1] if C.ndim == 2 and C.size else 0)  # C is n x m: one constraint per column
        nb = 0 if xl is None else len(xl)
        mt1 = m + nb + 1
        buf = np.zeros((cap, 2 * n + 2 * mt1 + 2))
        self.lib.orc_set_vector_trace(buf.ctypes.data_as(ct.c_void_p), cap)
        try:
            r = self.solve(G, a, C, bl, bu, xl, xu, **kw)
        finally:
            self.lib.orc_set_vector_trace(None, 0)
        out = []
        for rec in buf[: min(cap, r.iters)]:
            q = int(rec[2 * n + 2 * mt1 + 1])
            out.append(dict(x=rec[:n].copy(), u=rec[n : n + q + 1].copy(), z=rec[n + mt1 : 2 * n + mt1].copy(),
                            r=rec[2 * n + mt1 : 2 * n + mt1 + q].copy(), f=float(rec[2 * n + 2 * mt1]), q=q))
        return r, out

    def box_solve(self, x0, c, bl, xl, xu, max_iter=500, big_bnd=1e100, trace_cap=0) -> OracleResult:
        """experimental::BoxAndSingleConstraintSolver::solve(x0, c, bl, xl, xu): closed-form init_, then the GI loop."""
        x0, c, xl, xu = (np.ascontiguousarray(v, dtype=np.float64) for v in (x0, c, xl, xu))
        n = len(x0)
        x = np.zeros(n)
        u = np.zeros(1 + n)
        status = np.zeros(1 + n, dtype=np.int32)
        active = np.zeros(n + 2, dtype=np.int32)
        nact, iters, f = ct.c_int(0), ct.c_int(0), ct.c_double(0.0)
        tr = (TraceRec * max(trace_cap, 1))()
        term = self.lib.orc_box_solve(n, _dp(x0), _dp(c), float(bl), _dp(xl), _dp(xu), max_iter, big_bnd, _dp(x), _dp(u), _ip(status),
                                      _ip(active), ct.byref(nact), ct.byref(iters), ct.byref(f), tr, trace_cap)
        trace = [(t.p, t.status, t.l, t.kind, t.t1, t.t2) for t in tr[: min(trace_cap, iters.value)]]
        return OracleResult(term, x, u, status, active[: nact.value].copy(), iters.value, f.value, trace, None)

    def bench_rounds_fn(self, G, a, C, bl, bu, max_iter=500, big_bnd=1e100):
        """Timing loop of the CPU baseline (orc_gi_bench_rounds): returns run(rounds, nthreads) -> seconds of the rounds alone, and the
        output arrays of the last round.  Inputs are made contiguous ONCE, outputs are allocated ONCE; G is restored per solve by a
        per-thread memcpy inside the C loop (benchmarks/Solvers.cpp:451), nothing is copied or allocated in Python per round."""
        B, n, _ = G.shape
        m = C.shape[1]
        G, a, C, bl, bu = (np.ascontiguousarray(v, dtype=np.float64) for v in (G, a, C, bl, bu))
        out = dict(x=np.zeros((B, n)), u=np.zeros((B, m)), status=np.zeros((B, m), dtype=np.int32), term=np.zeros(B, dtype=np.int32),
                   iters=np.zeros(B, dtype=np.int32), f=np.zeros(B))

        def run(rounds, nthreads):
            return float(self.lib.orc_gi_bench_rounds(rounds, B, n, m, 0, _dp(G), _dp(a), _dp(C), _dp(bl), _dp(bu), None, None, max_iter, big_bnd,
                                                      _dp(out["x"]), _dp(out["u"]), _ip(out["status"]), _ip(out["term"]), _ip(out["iters"]),
                                                      _dp(out["f"]), nthreads))

        return run, out

    def solve_batch(self, G, a, C, bl, bu, xl=None, xu=None, max_iter=500, big_bnd=1e100, nthreads=1, warm=False, guess=None):
        """Packed batch (see jrl-qp_amd/problems.py PackedBatch).  G is copied.  warm/guess: see solve()."""
        B, n, _ = G.shape
        m = C.shape[1]
        nb = 0 if xl is None or xl.shape[1] == 0 else xl.shape[1]
        Gc = np.ascontiguousarray(G, dtype=np.float64).copy()
        a = np.ascontiguousarray(a)
        C = np.ascontiguousarray(C)
        bl = np.ascontiguousarray(bl)
        bu = np.ascontiguousarray(bu)
        xlc = np.ascontiguousarray(xl) if nb else None
        xuc = np.ascontiguousarray(xu) if nb else None
        x = np.zeros((B, n))
        u = np.zeros((B, m + nb))
        status = np.zeros((B, m + nb), dtype=np.int32)
        term = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        f = np.zeros(B)
        if warm:
            g = None if guess is None else np.ascontiguousarray(guess, dtype=np.int32)
            assert g is None or g.shape == (B, m + nb)
            self.lib.orc_gi_solve_batch_warm(B, n, m, nb, _dp(Gc), _dp(a), _dp(C), _dp(bl), _dp(bu), _dp(xlc), _dp(xuc), _ip(g),
                                             max_iter, big_bnd, _dp(x), _dp(u), _ip(status), _ip(term), _ip(iters), _dp(f), nthreads)
        else:
            self.lib.orc_gi_solve_batch(B, n, m, nb, _dp(Gc), _dp(a), _dp(C), _dp(bl), _dp(bu), _dp(xlc), _dp(xuc), max_iter, big_bnd,
                                        _dp(x), _dp(u), _ip(status), _ip(term), _ip(iters), _dp(f), nthreads)
        return dict(x=x, u=u, status=status, term=term, iters=iters, f=f, L=Gc)

    LIT_CHAIN_DOTS, LIT_ROT_C3, LIT_SERIAL_GIVENS, LIT_NO_FMA, LIT_CHAIN_GEMV, LIT_R_SOLVE, LIT_ALL = 1, 2, 4, 8, 16, 32, 55

    def solve_batch_literal(self, flags, G, a, C, bl, bu, xl=None, xu=None, max_iter=500, big_bnd=1e100, nthreads=1):
        """Cold packed batch with some of CANON C5 / C6 / C7 switched off (orc_gi_solve_batch_literal; flags = 0 is the canonical
        solve).  Also returns the ordered active lists (B x n, -1 beyond nactive)."""
        B, n, _ = G.shape
        m = C.shape[1]
        nb = 0 if xl is None or xl.shape[1] == 0 else xl.shape[1]
        Gc = np.ascontiguousarray(G, dtype=np.float64).copy()
        a, C, bl, bu = (np.ascontiguousarray(v, dtype=np.float64) for v in (a, C, bl, bu))
        xlc = np.ascontiguousarray(xl) if nb else None
        xuc = np.ascontiguousarray(xu) if nb else None
        out = dict(x=np.zeros((B, n)), u=np.zeros((B, m + nb)), status=np.zeros((B, m + nb), dtype=np.int32),
                   active=np.zeros((B, n), dtype=np.int32), nactive=np.zeros(B, dtype=np.int32), term=np.zeros(B, dtype=np.int32),
                   iters=np.zeros(B, dtype=np.int32), f=np.zeros(B))
        self.lib.orc_gi_solve_batch_literal(flags, B, n, m, nb, _dp(Gc), _dp(a), _dp(C), _dp(bl), _dp(bu), _dp(xlc), _dp(xuc), max_iter,
                                            big_bnd, _dp(out["x"]), _dp(out["u"]), _ip(out["status"]), _ip(out["active"]),
                                            _ip(out["nactive"]), _ip(out["term"]), _ip(out["iters"]), _dp(out["f"]), nthreads)
        return out

    # ---- primitives ----------------------------------------------------------------------------
    def llt(self, A):
        Af = np.array(A, dtype=np.float64, order="F", copy=True)
        ret = self.lib.orc_llt(Af.shape[0], _dp(Af), max(Af.shape[0], 1))
        return ret, Af

    def inv_lt(self, L):
        Lf = np.asfortranarray(L, dtype=np.float64)
        n = Lf.shape[0]
        J = np.zeros((n, n), order="F")
        self.lib.orc_inv_lt(n, _dp(Lf), n, _dp(J), n)
        return J

    def givens(self, p, q):
        c, s, r = ct.c_double(), ct.c_double(), ct.c_double()
        self.lib.orc_givens(p, q, ct.byref(c), ct.byref(s), ct.byref(r))
        return c.value, s.value, r.value

    def dot(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.lib.orc_dot(len(a), _dp(a), _dp(b))

    def dot_blk(self, a, b, lo=0):
        """CANON C8: the C1 chain up to 256 entries, 64-term segments + the C4 tree beyond."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.lib.orc_dot_blk(len(a), _dp(a), _dp(b), lo)

    # ---- structured ----------------------------------------------------------------------------
    @staticmethod
    def _blocks(blocks):
        """blocks: list of Fortran-ordered 2-D views (possibly into one big matrix)."""
        k = len(blocks)
        ptrs = (c_dp * max(k, 1))()
        lds = np.zeros(max(k, 1), dtype=np.int32)
        for i, b in enumerate(blocks):
            if b.size == 0:
                lds[i] = max(b.shape[0], 1)
                continue
            assert b.dtype == np.float64 and b.strides[0] == 8, "blocks must be col-major views"
            ptrs[i] = ct.cast(b.ctypes.data, c_dp)
            lds[i] = b.strides[1] // 8 if b.shape[1] > 1 else max(b.shape[0], 1)
        return ptrs, lds

    def block_arrow_llt(self, diag, side, up=False):
        bs = np.array([d.shape[0] for d in diag], dtype=np.int32)
        dp, dl = self._blocks(diag)
        sp, sl = self._blocks(side)
        return bool(self.lib.orc_block_arrow_llt(len(diag), _ip(bs), dp, _ip(dl), sp, _ip(sl), int(up)))

    def block_arrow_solve(self, diag, side, up, M, start=0, end=-1, transpose=False):
        bs = np.array([d.shape[0] for d in diag], dtype=np.int32)
        dp, dl = self._blocks(diag)
        sp, sl = self._blocks(side)
        assert M.flags.f_contiguous
        fn = self.lib.orc_block_arrow_ltsolve if transpose else self.lib.orc_block_arrow_lsolve
        fn(len(diag), _ip(bs), dp, _ip(dl), sp, _ip(sl), int(up), _dp(M), M.shape[0], M.shape[1], start, end)

    def tri_block_diag_llt(self, diag, sub):
        bs = np.array([d.shape[0] for d in diag], dtype=np.int32)
        dp, dl = self._blocks(diag)
        sp, sl = self._blocks(sub)
        return bool(self.lib.orc_tri_block_diag_llt(len(diag), _ip(bs), dp, _ip(dl), sp, _ip(sl)))

    def tri_block_diag_solve(self, diag, sub, M, start=0, end=-1, transpose=False):
        bs = np.array([d.shape[0] for d in diag], dtype=np.int32)
        dp, dl = self._blocks(diag)
        sp, sl = self._blocks(sub)
        assert M.flags.f_contiguous
        if transpose:
            self.lib.orc_tri_block_diag_ltsolve(len(diag), _ip(bs), dp, _ip(dl), sp, _ip(sl), _dp(M), M.shape[0], M.shape[1], end)
        else:
            self.lib.orc_tri_block_diag_lsolve(len(diag), _ip(bs), dp, _ip(dl), sp, _ip(sl), _dp(M), M.shape[0], M.shape[1], start)


    # ---- structured GI loop (blockgi_oracle.c) ---------------------------------------------------
    def blockgi_solve(self, kind, diag, off, a, Cblocks, bl, bu, xl=None, xu=None, max_iter=500, big_bnd=1e100, trace_cap=0, c9=False):
        """experimental::BlockGISolver::solve; c9=True: contract C9 (no orthonormal sequence: sparse B = L^-1 N and S = R^-1, orc_blockgi_solve_c9).  kind: "tri" | "arrow_up" | "arrow_down"; diag/off: col-major views
        (factorised IN PLACE); Cblocks: list of n_i x m_i col-major blocks of the block-diagonal C."""
        typ = {"tri": 0, "arrow_up": 1, "arrow_down": 2}[kind]
        bs = np.array([d.shape[0] for d in diag], dtype=np.int32)
        n = int(bs.sum())
        dp, dl = self._blocks(diag)
        op, ol = self._blocks(off)
        cv = np.array([c.shape[0] for c in Cblocks], dtype=np.int32)
        cc = np.array([c.shape[1] for c in Cblocks], dtype=np.int32)
        Cb = [np.asfortranarray(c, dtype=np.float64) for c in Cblocks]
        cp, cl = self._blocks(Cb)
        m = int(cc.sum())
        a = np.ascontiguousarray(a, dtype=np.float64)
        bl = np.ascontiguousarray(bl, dtype=np.float64)
        bu = np.ascontiguousarray(bu, dtype=np.float64)
        nb = 0 if xl is None else len(xl)
        xl = np.ascontiguousarray(xl, dtype=np.float64) if nb else None
        xu = np.ascontiguousarray(xu, dtype=np.float64) if nb else None
        x = np.zeros(n)
        u = np.zeros(m + nb)
        status = np.zeros(m + nb, dtype=np.int32)
        active = np.zeros(m + nb + 1, dtype=np.int32)
        nact, iters, f = ct.c_int(0), ct.c_int(0), ct.c_double(0.0)
        tr = (TraceRec * max(trace_cap, 1))()
        fn = self.lib.orc_blockgi_solve_c9 if c9 else self.lib.orc_blockgi_solve
        term = fn(typ, len(diag), _ip(bs), dp, _ip(dl), op, _ip(ol), _dp(a), len(Cb), _ip(cv), _ip(cc), cp,
                                          _ip(cl), _dp(bl), _dp(bu), _dp(xl), _dp(xu), nb, max_iter, big_bnd, _dp(x), _dp(u),
                                          _ip(status), _ip(active), ct.byref(nact), ct.byref(iters), ct.byref(f), tr, trace_cap)
        trace = [(t.p, t.status, t.l, t.kind, t.t1, t.t2) for t in tr[: min(trace_cap, iters.value)]]
        return OracleResult(term, x, u, status, active[: nact.value].copy(), iters.value, f.value, trace, None)

    def dot_tree(self, a, b, phase=0):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        return self.lib.orc_dot_tree(len(a), _dp(a), _dp(b), phase)


class OrthoSeq:
    """OrthonormalSequence of the oracle (for the identities of the reference's tests/InternalTest.cpp)."""

    def __init__(self, oracle, n):
        self.lib, self.n = oracle.lib, n
        self.h = self.lib.orc_oseq_new(n)

    def add_householder(self, start, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        return self.lib.orc_oseq_add_householder(self.h, start, len(v), _dp(v))

    def add_givens(self, start, c, s):
        c = np.ascontiguousarray(c, dtype=np.float64)
        s = np.ascontiguousarray(s, dtype=np.float64)
        self.lib.orc_oseq_add_givens(self.h, start, len(c), _dp(c), _dp(s))

    def apply(self, v, transpose=False):
        w = np.array(v, dtype=np.float64).copy()
        assert w.shape == (self.n,)
        self.lib.orc_oseq_apply(self.h, _dp(w), int(transpose))
        return w

    def to_dense(self):
        return np.stack([self.apply(e) for e in np.eye(self.n)], axis=1)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_oseq_free(self.h)
            self.h = None


_ORACLE = None


def get_oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
