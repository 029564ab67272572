"""ctypes binding of libjrlqp_hip.so (the C ABI of include/jrlqp_hip.h) + torch plumbing for device memory.

Python is NOT on the numerical path: every function here only marshals device pointers into the C ABI.
If the HIP library is missing or cannot be loaded the import of this module's users fails loudly -- there is
no CPU fallback anywhere in the product.
"""
from __future__ import annotations

import ctypes as ct
import os
from dataclasses import dataclass

import numpy as np

from . import LIB_PATH

c_dp = ct.c_void_p

# every symbol include/jrlqp_hip.h declares (tests/test_capi_symbols.py checks the list against the header)
EXPORTED = [
    "jrlqp_abi_version",
    "jrlqp_status_string",
    "jrlqp_last_hip_error",
    "jrlqp_device_count",
    "jrlqp_default_options",
    "jrlqp_gi_solve_batch",
    "jrlqp_gi_batch_lds_bytes",
    "jrlqp_debug_set_profile",
    "jrlqp_selftest_arith",
    "jrlqp_selftest_mfma_f64",
]

TERMINATION = ["SUCCESS", "INCONSISTENT_INPUT", "NON_POS_HESSIAN", "INFEASIBLE", "MAX_ITER_REACHED",
               "LINEAR_DEPENDENCY_DETECTED", "OVERCONSTRAINED_PROBLEM", "UNKNOWN"]


class Options(ct.Structure):
    _fields_ = [("max_iter", ct.c_int), ("big_bnd", ct.c_double)]


class TraceRec(ct.Structure):
    _fields_ = [("p", ct.c_int), ("status", ct.c_int), ("l", ct.c_int), ("kind", ct.c_int), ("t1", ct.c_double), ("t2", ct.c_double)]


class JrlqpError(RuntimeError):
    pass


_LIB = None


def load_library() -> ct.CDLL:
    """dlopen the HIP library and bind every exported symbol.  Raises if it is missing: build it with
    `python -c "import __graft_entry__ as g; g.build()"` (or `make -C jrl-qp_amd/csrc`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise JrlqpError(f"{LIB_PATH} not found: the HIP extension is not built (no CPU fallback exists)")
    lib = ct.CDLL(LIB_PATH)
    for name in EXPORTED:
        if not hasattr(lib, name):
            raise JrlqpError(f"{LIB_PATH} does not export {name}")
    lib.jrlqp_abi_version.restype = ct.c_int
    lib.jrlqp_status_string.restype = ct.c_char_p
    lib.jrlqp_status_string.argtypes = [ct.c_int]
    lib.jrlqp_last_hip_error.restype = ct.c_int
    lib.jrlqp_device_count.restype = ct.c_int
    lib.jrlqp_default_options.restype = None
    lib.jrlqp_default_options.argtypes = [ct.POINTER(Options)]
    lib.jrlqp_gi_batch_lds_bytes.restype = ct.c_long
    lib.jrlqp_gi_batch_lds_bytes.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_int)]
    lib.jrlqp_gi_solve_batch.restype = ct.c_int
    lib.jrlqp_gi_solve_batch.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int] + [c_dp] * 7 + [ct.POINTER(Options)] + [c_dp] * 8 + [ct.c_int, c_dp]
    lib.jrlqp_debug_set_profile.restype = ct.c_int
    lib.jrlqp_debug_set_profile.argtypes = [c_dp]
    lib.jrlqp_selftest_arith.restype = ct.c_int
    lib.jrlqp_selftest_arith.argtypes = [ct.c_int] + [c_dp] * 9 + [c_dp]
    lib.jrlqp_selftest_mfma_f64.restype = ct.c_int
    lib.jrlqp_selftest_mfma_f64.argtypes = [c_dp] * 4 + [c_dp]
    _LIB = lib
    return lib


def check(status: int):
    if status != 0:
        lib = load_library()
        msg = lib.jrlqp_status_string(status).decode()
        raise JrlqpError(f"jrlqp error {status}: {msg} (hip error {lib.jrlqp_last_hip_error()})")


def _ptr(t):
    return ct.c_void_p(t.data_ptr()) if t is not None else None


@dataclass
class BatchResult:
    x: "torch.Tensor"  # (B, n) f64
    u: "torch.Tensor"  # (B, m+nb) f64, sign-expanded multipliers
    status: "torch.Tensor"  # (B, m+nb) int32 activation status
    term: "torch.Tensor"  # (B,) int32 TerminationStatus
    iters: "torch.Tensor"  # (B,) int32
    f: "torch.Tensor"  # (B,) f64
    L: "torch.Tensor | None" = None
    trace: "torch.Tensor | None" = None


class BatchSolver:
    """Pre-sized batched solver: owns the output buffers (like the reference solver owns work_x_/work_u_,
    DualSolver.h:128-141), so `solve` performs no allocation and no host synchronisation."""

    def __init__(self, n: int, m: int, use_bounds: bool, B: int, device="cuda", max_iter: int = 500, big_bnd: float = 1e100,
                 want_factor: bool = False, trace_cap: int = 0):
        import torch

        self.torch = torch
        self.lib = load_library()
        self.n, self.m, self.nb, self.B = n, m, (n if use_bounds else 0), B
        self.device = torch.device(device)
        c = ct.c_int(0)
        lds = self.lib.jrlqp_gi_batch_lds_bytes(n, m, self.nb, ct.byref(c))
        if lds < 0:
            raise JrlqpError(f"batched path does not support n={n}, m={m}")
        self.lds_bytes, self.c_placement = int(lds), int(c.value)  # 2 registers, 1 LDS, 0 streamed
        self.opts = Options(max_iter, big_bnd)
        mt = m + self.nb
        f64, i32 = torch.float64, torch.int32
        self.x = torch.empty((B, n), dtype=f64, device=self.device)
        self.u = torch.empty((B, mt), dtype=f64, device=self.device)
        self.status = torch.empty((B, mt), dtype=i32, device=self.device)
        self.term = torch.empty((B,), dtype=i32, device=self.device)
        self.iters = torch.empty((B,), dtype=i32, device=self.device)
        self.f = torch.empty((B,), dtype=f64, device=self.device)
        self.L = torch.zeros((B, n, n), dtype=f64, device=self.device) if want_factor else None
        self.trace_cap = trace_cap
        self.trace = torch.zeros((B, trace_cap, ct.sizeof(TraceRec)), dtype=torch.uint8, device=self.device) if trace_cap else None

    def solve(self, G, a, C, bl, bu, xl=None, xu=None, stream=None) -> BatchResult:
        """All inputs: contiguous float64 CUDA tensors in the packed layout of include/jrlqp_hip.h
        (G (B,n,n), a (B,n), C (B,m,n) i.e. n x m col-major per problem, bl/bu (B,m), xl/xu (B,nb))."""
        torch = self.torch
        for t in (G, a, C, bl, bu, xl, xu):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        assert G.shape == (self.B, self.n, self.n) and a.shape == (self.B, self.n)
        assert C.shape == (self.B, self.m, self.n) and bl.shape == (self.B, self.m) and bu.shape == (self.B, self.m)
        if self.nb:
            assert xl.shape == (self.B, self.nb) and xu.shape == (self.B, self.nb)
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        st = self.lib.jrlqp_gi_solve_batch(
            self.n, self.m, self.nb, self.B, _ptr(G), _ptr(a), _ptr(C), _ptr(bl), _ptr(bu),
            _ptr(xl) if self.nb else None, _ptr(xu) if self.nb else None, ct.byref(self.opts),
            _ptr(self.x), _ptr(self.u), _ptr(self.status), _ptr(self.term), _ptr(self.iters), _ptr(self.f),
            _ptr(self.L), _ptr(self.trace), self.trace_cap, ct.c_void_p(s.cuda_stream))
        check(st)
        return BatchResult(self.x, self.u, self.status, self.term, self.iters, self.f, self.L, self.trace)

    def trace_numpy(self):
        """Decode the trace buffer -> structured array (B, trace_cap)."""
        raw = self.trace.cpu().numpy()
        dt = np.dtype([("p", "<i4"), ("status", "<i4"), ("l", "<i4"), ("kind", "<i4"), ("t1", "<f8"), ("t2", "<f8")])
        return raw.view(dt).reshape(self.B, self.trace_cap)


def selftest_arith(a, b, c):
    """Device a/b, sqrt|a|, fma(a,b,c) and makeGivens(a,b) for float64 CUDA tensors a, b, c."""
    import torch

    lib = load_library()
    outs = [torch.empty_like(a) for _ in range(6)]
    check(lib.jrlqp_selftest_arith(a.numel(), _ptr(a), _ptr(b), _ptr(c), *[_ptr(o) for o in outs],
                                   ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return outs


def selftest_mfma_f64(A, B, C):
    """D = A(16x4) B(4x16) + C(16x16) by ONE v_mfma_f64_16x16x4_f64 (row-major float64 CUDA tensors)."""
    import torch

    lib = load_library()
    D = torch.empty((16, 16), dtype=torch.float64, device=A.device)
    check(lib.jrlqp_selftest_mfma_f64(_ptr(A), _ptr(B), _ptr(C), _ptr(D), ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return D


def to_device(batch, device="cuda"):
    """PackedBatch (numpy, problems.py) -> dict of contiguous CUDA tensors."""
    import torch

    d = {}
    for k in ("G", "a", "C", "bl", "bu", "xl", "xu"):
        d[k] = torch.from_numpy(np.ascontiguousarray(getattr(batch, k))).to(device)
    return d


def smoke_check(oracle):
    """One small batched solve on cuda:0, checked bit for bit against the CPU oracle (the checker)."""
    import torch

    from . import problems

    if not torch.cuda.is_available():
        raise JrlqpError("smoke() needs a GPU")
    ch = problems.benchmark_characteristics(10, 20)
    pb = problems.generate_batch(64, ch)
    dev = to_device(pb, "cuda:0")
    solver = BatchSolver(pb.n, pb.m, False, pb.B, device="cuda:0")
    r = solver.solve(dev["G"], dev["a"], dev["C"], dev["bl"], dev["bu"])
    torch.cuda.synchronize()
    ref = oracle.solve_batch(pb.G, pb.a, pb.C, pb.bl, pb.bu)
    assert np.array_equal(r.term.cpu().numpy(), ref["term"]), "termination status differs from the oracle"
    assert np.array_equal(r.iters.cpu().numpy(), ref["iters"]), "iteration counts differ from the oracle"
    assert np.array_equal(r.status.cpu().numpy(), ref["status"]), "active sets differ from the oracle"
    assert np.array_equal(r.x.cpu().numpy(), ref["x"]), "x is not bit-identical to the oracle"
    assert np.array_equal(r.u.cpu().numpy(), ref["u"]), "u is not bit-identical to the oracle"
    assert np.array_equal(r.f.cpu().numpy(), ref["f"]), "f is not bit-identical to the oracle"
    print(f"smoke ok: {pb.B} QPs (n={pb.n}, m={pb.m}) bit-identical to the oracle; mean iterations {ref['iters'].mean():.2f}")
