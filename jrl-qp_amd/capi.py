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
    "jrlqp_gi_solve_batch_warm",
    "jrlqp_gi_batch_workspace_bytes",
    "jrlqp_gi_solve_batch_ws",
    "jrlqp_box_solve_batch",
    "jrlqp_box_solve_batch_ws",
    "jrlqp_shard_range",
    "jrlqp_gi_batch_workspace_bytes_on",
    "jrlqp_gi_solve_batch_multi",
    "jrlqp_gi_gather_x_multi",
    "jrlqp_gi_batch_lds_bytes",
    "jrlqp_gi_batch_warm_lds_bytes",
    "jrlqp_dense_create",
    "jrlqp_dense_destroy",
    "jrlqp_dense_resize",
    "jrlqp_dense_last_hip_error",
    "jrlqp_dense_bind",
    "jrlqp_dense_init",
    "jrlqp_dense_init_active_set",
    "jrlqp_dense_select",
    "jrlqp_dense_step",
    "jrlqp_dense_advance",
    "jrlqp_dense_add",
    "jrlqp_dense_remove",
    "jrlqp_dense_solve",
    "jrlqp_dense_solve_warm",
    "jrlqp_dense_results",
    "jrlqp_dense_set_trace",
    "jrlqp_dense_sync_mode",
    "jrlqp_blocks_create",
    "jrlqp_blocks_destroy",
    "jrlqp_blocks_bind",
    "jrlqp_blocks_llt",
    "jrlqp_blocks_result",
    "jrlqp_blocks_solve",
    "jrlqp_gi_multi_failed_shard",
    "jrlqp_debug_multi_allow_shared_device",
    "jrlqp_debug_multi_inject_failure",
    "jrlqp_dense_trace_vectors_stride",
    "jrlqp_dense_set_trace_vectors",
    "jrlqp_blockgi_create",
    "jrlqp_blockgi_create_form",
    "jrlqp_blockgi_form",
    "jrlqp_blockgi_destroy",
    "jrlqp_blockgi_workspace_bytes",
    "jrlqp_blockgi_bind",
    "jrlqp_blockgi_solve",
    "jrlqp_debug_set_profile",
    "jrlqp_debug_dense_inject_sync_fault",
    "jrlqp_selftest_arith",
    "jrlqp_selftest_mfma_f64",
]

TERMINATION = ["SUCCESS", "INCONSISTENT_INPUT", "NON_POS_HESSIAN", "INFEASIBLE", "MAX_ITER_REACHED",
               "LINEAR_DEPENDENCY_DETECTED", "OVERCONSTRAINED_PROBLEM", "UNKNOWN"]


class Options(ct.Structure):
    _fields_ = [("max_iter", ct.c_int), ("big_bnd", ct.c_double)]


class TraceRec(ct.Structure):
    _fields_ = [("p", ct.c_int), ("status", ct.c_int), ("l", ct.c_int), ("kind", ct.c_int), ("t1", ct.c_double), ("t2", ct.c_double)]


class ShardBuffers(ct.Structure):
    """struct jrlqp_shard_buffers (include/jrlqp_hip.h): the buffers of one device's shard of a multi-device batch."""

    _fields_ = [(k, ct.c_void_p) for k in ("dG", "da", "dC", "dbl", "dbu", "dxl", "dxu", "das", "dx", "du", "dstatus", "dterm", "diters", "df", "dL",
                                           "dworkspace")] + [("workspace_bytes", ct.c_longlong), ("stream", ct.c_void_p)]


class JrlqpError(RuntimeError):
    """An error code of the ABI itself (enum jrlqp_status); `status` is that code (0 when the library could not be loaded)."""

    def __init__(self, msg, status=0):
        super().__init__(msg)
        self.status = status


# enum jrlqp_status (include/jrlqp_hip.h)
ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED_SIZE, ERR_NO_DEVICE, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4, -5


_LIB = None


def load_library() -> ct.CDLL:
    """dlopen the HIP library and bind every exported symbol.  Raises if it is missing: build it with
    `python -c "import __graft_entry__ as g; g.build()"` (or `make -C jrl-qp_amd/csrc`)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise JrlqpError(f"{LIB_PATH} not found: the HIP extension is not built (no CPU fallback exists)")
    # torch first: the process must end up with ONE HIP runtime.  torch ships its own libamdhip64; if this library were
    # dlopen'ed before torch, it would bind the system runtime and every launch on torch-allocated memory would fail
    # (hipErrorNoDevice) -- torch is what the callers use for device memory and streams anyway.
    import torch  # noqa: F401

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
    lib.jrlqp_gi_batch_warm_lds_bytes.restype = ct.c_long
    lib.jrlqp_gi_batch_warm_lds_bytes.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_int)]
    lib.jrlqp_gi_solve_batch_warm.restype = ct.c_int
    lib.jrlqp_gi_solve_batch_warm.argtypes = ([ct.c_int, ct.c_int, ct.c_int, ct.c_int] + [c_dp] * 8 + [ct.POINTER(Options)] + [c_dp] * 8
                                              + [ct.c_int, c_dp])
    lib.jrlqp_gi_batch_workspace_bytes.restype = ct.c_longlong
    lib.jrlqp_gi_batch_workspace_bytes.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int]
    lib.jrlqp_gi_solve_batch_ws.restype = ct.c_int
    lib.jrlqp_gi_solve_batch_ws.argtypes = ([ct.c_int] * 5 + [c_dp] * 8 + [ct.POINTER(Options)] + [c_dp] * 8
                                            + [ct.c_int, c_dp, ct.c_longlong, c_dp])
    lib.jrlqp_box_solve_batch.restype = ct.c_int
    lib.jrlqp_box_solve_batch.argtypes = [ct.c_int, ct.c_int] + [c_dp] * 5 + [ct.POINTER(Options)] + [c_dp] * 6 + [c_dp, ct.c_int, c_dp]
    lib.jrlqp_box_solve_batch_ws.restype = ct.c_int
    lib.jrlqp_box_solve_batch_ws.argtypes = [ct.c_int, ct.c_int] + [c_dp] * 5 + [ct.POINTER(Options)] + [c_dp] * 6 + [c_dp, ct.c_int, c_dp, ct.c_longlong, c_dp]
    lib.jrlqp_gi_solve_batch.restype = ct.c_int
    lib.jrlqp_gi_solve_batch.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.c_int] + [c_dp] * 7 + [ct.POINTER(Options)] + [c_dp] * 8 + [ct.c_int, c_dp]
    lib.jrlqp_dense_create.restype = ct.c_int
    lib.jrlqp_dense_create.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_void_p)]
    lib.jrlqp_dense_destroy.restype = None
    lib.jrlqp_dense_destroy.argtypes = [ct.c_void_p]
    lib.jrlqp_dense_resize.restype = ct.c_int
    lib.jrlqp_dense_resize.argtypes = [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int]
    lib.jrlqp_dense_last_hip_error.restype = ct.c_int
    lib.jrlqp_dense_last_hip_error.argtypes = [ct.c_void_p]
    lib.jrlqp_dense_solve.restype = ct.c_int
    lib.jrlqp_dense_solve.argtypes = [ct.c_void_p, c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_dp, c_dp, c_dp, c_dp, ct.POINTER(Options), ct.POINTER(ct.c_int), c_dp]
    lib.jrlqp_dense_solve_warm.restype = ct.c_int
    lib.jrlqp_dense_solve_warm.argtypes = [ct.c_void_p, c_dp, ct.c_int, c_dp, c_dp, ct.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, ct.POINTER(Options),
                                           ct.POINTER(ct.c_int), c_dp]
    lib.jrlqp_dense_results.restype = ct.c_int
    lib.jrlqp_dense_results.argtypes = [ct.c_void_p, c_dp, c_dp, c_dp, ct.POINTER(ct.c_int), ct.POINTER(ct.c_double), c_dp]
    lib.jrlqp_blocks_create.restype = ct.c_int
    lib.jrlqp_blocks_create.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_int), ct.c_int, ct.c_int, ct.POINTER(ct.c_void_p)]
    lib.jrlqp_blocks_destroy.restype = None
    lib.jrlqp_blocks_destroy.argtypes = [ct.c_void_p]
    lib.jrlqp_blocks_bind.restype = ct.c_int
    lib.jrlqp_blocks_bind.argtypes = [ct.c_void_p, ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_int), ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_int), c_dp]
    lib.jrlqp_blocks_llt.restype = ct.c_int
    lib.jrlqp_blocks_llt.argtypes = [ct.c_void_p, c_dp]
    lib.jrlqp_blocks_result.restype = ct.c_int
    lib.jrlqp_blocks_result.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int), c_dp]
    lib.jrlqp_blocks_solve.restype = ct.c_int
    lib.jrlqp_blocks_solve.argtypes = [ct.c_void_p, ct.c_int, c_dp, ct.c_int, ct.c_long, ct.c_int, ct.c_int, ct.c_int, c_dp]
    lib.jrlqp_blockgi_create.restype = ct.c_int
    lib.jrlqp_blockgi_create.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_int), ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int),
                                         ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_void_p)]
    lib.jrlqp_blockgi_create_form.restype = ct.c_int
    lib.jrlqp_blockgi_create_form.argtypes = [ct.c_int, ct.c_int, ct.POINTER(ct.c_int), ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int),
                                              ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_void_p)]
    lib.jrlqp_blockgi_form.restype = ct.c_int
    lib.jrlqp_blockgi_form.argtypes = [ct.c_void_p]
    lib.jrlqp_blockgi_destroy.restype = None
    lib.jrlqp_blockgi_destroy.argtypes = [ct.c_void_p]
    lib.jrlqp_blockgi_workspace_bytes.restype = ct.c_longlong
    lib.jrlqp_blockgi_workspace_bytes.argtypes = [ct.c_void_p]
    lib.jrlqp_blockgi_bind.restype = ct.c_int
    lib.jrlqp_blockgi_bind.argtypes = [ct.c_void_p, ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_int), ct.POINTER(ct.c_void_p),
                                       ct.POINTER(ct.c_int), ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_int), c_dp]
    lib.jrlqp_blockgi_solve.restype = ct.c_int
    lib.jrlqp_blockgi_solve.argtypes = [ct.c_void_p] + [c_dp] * 5 + [ct.POINTER(Options)] + [c_dp] * 6 + [c_dp, ct.c_int, c_dp]
    lib.jrlqp_shard_range.restype = None
    lib.jrlqp_shard_range.argtypes = [ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
    lib.jrlqp_gi_batch_workspace_bytes_on.restype = ct.c_longlong
    lib.jrlqp_gi_batch_workspace_bytes_on.argtypes = [ct.c_int] * 5
    lib.jrlqp_gi_solve_batch_multi.restype = ct.c_int
    lib.jrlqp_gi_solve_batch_multi.argtypes = [ct.c_int] * 6 + [ct.POINTER(ct.c_int), ct.POINTER(ShardBuffers), ct.POINTER(Options), ct.c_int]
    lib.jrlqp_gi_gather_x_multi.restype = ct.c_int
    lib.jrlqp_gi_gather_x_multi.argtypes = [ct.c_int] * 3 + [ct.POINTER(ct.c_int), ct.POINTER(ShardBuffers), ct.c_int, ct.c_void_p, ct.c_int]
    lib.jrlqp_debug_set_profile.restype = ct.c_int
    lib.jrlqp_debug_dense_inject_sync_fault.restype = ct.c_int
    lib.jrlqp_debug_dense_inject_sync_fault.argtypes = [ct.c_void_p, ct.c_int]
    lib.jrlqp_dense_set_trace.restype = ct.c_int
    lib.jrlqp_dense_set_trace.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int]
    lib.jrlqp_dense_sync_mode.restype = ct.c_int
    lib.jrlqp_dense_sync_mode.argtypes = [ct.c_void_p, ct.POINTER(ct.c_int)]
    lib.jrlqp_dense_init_active_set.restype = ct.c_int
    lib.jrlqp_dense_init_active_set.argtypes = [ct.c_void_p, ct.c_double, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int), ct.c_void_p, ct.c_void_p,
                                                ct.POINTER(ct.c_double), ct.c_void_p]
    lib.jrlqp_debug_set_profile.argtypes = [ct.c_int, c_dp, ct.c_int, ct.c_int]
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
        raise JrlqpError(f"jrlqp error {status}: {msg} (hip error {lib.jrlqp_last_hip_error()})", status)


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
        # the workspace of a pre-sized solver (jrlqp_gi_batch_workspace_bytes): allocated HERE, never inside solve()
        with torch.cuda.device(self.device):
            wsb = int(self.lib.jrlqp_gi_batch_workspace_bytes(n, m, self.nb, B))
        if wsb < 0:
            raise JrlqpError(f"batched path does not support n={n}, m={m}")
        self.workspace_bytes = wsb
        self.workspace = torch.empty((wsb + 7) // 8, dtype=f64, device=self.device) if wsb else None

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
        st = self.lib.jrlqp_gi_solve_batch_ws(
            0, self.n, self.m, self.nb, self.B, _ptr(G), _ptr(a), _ptr(C), _ptr(bl), _ptr(bu),
            _ptr(xl) if self.nb else None, _ptr(xu) if self.nb else None, None, ct.byref(self.opts),
            _ptr(self.x), _ptr(self.u), _ptr(self.status), _ptr(self.term), _ptr(self.iters), _ptr(self.f),
            _ptr(self.L), _ptr(self.trace), self.trace_cap, _ptr(self.workspace), self.workspace_bytes, ct.c_void_p(s.cuda_stream))
        check(st)
        return BatchResult(self.x, self.u, self.status, self.term, self.iters, self.f, self.L, self.trace)

    def solve_warm(self, G, a, C, bl, bu, xl=None, xu=None, guess=None, stream=None) -> BatchResult:
        """Warm-started solve (experimental::GoldfarbIdnaniSolver with warmStart_, jrlqp_gi_solve_batch_warm).
        guess: (B, m+nb) int32 CUDA tensor of activation statuses, "previous" to reuse this solver's own last
        active set in place (the reference's as.empty() case), or None for no guess."""
        torch = self.torch
        for t in (G, a, C, bl, bu, xl, xu):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        assert G.shape == (self.B, self.n, self.n) and a.shape == (self.B, self.n)
        assert C.shape == (self.B, self.m, self.n) and bl.shape == (self.B, self.m) and bu.shape == (self.B, self.m)
        if self.nb:
            assert xl.shape == (self.B, self.nb) and xu.shape == (self.B, self.nb)
        if isinstance(guess, str):
            assert guess == "previous"
            guess = self.status
        if guess is not None:
            assert guess.is_cuda and guess.dtype == torch.int32 and guess.is_contiguous() and guess.shape == (self.B, self.m + self.nb)
        c = ct.c_int(0)
        if self.lib.jrlqp_gi_batch_warm_lds_bytes(self.n, self.m, self.nb, ct.byref(c)) < 0:
            raise JrlqpError(f"warm-started batched path does not support n={self.n}, m={self.m}")
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        st = self.lib.jrlqp_gi_solve_batch_ws(
            1, self.n, self.m, self.nb, self.B, _ptr(G), _ptr(a), _ptr(C), _ptr(bl), _ptr(bu),
            _ptr(xl) if self.nb else None, _ptr(xu) if self.nb else None, _ptr(guess), ct.byref(self.opts),
            _ptr(self.x), _ptr(self.u), _ptr(self.status), _ptr(self.term), _ptr(self.iters), _ptr(self.f),
            _ptr(self.L), _ptr(self.trace), self.trace_cap, _ptr(self.workspace), self.workspace_bytes, ct.c_void_p(s.cuda_stream))
        check(st)
        return BatchResult(self.x, self.u, self.status, self.term, self.iters, self.f, self.L, self.trace)

    def trace_numpy(self):
        """Decode the trace buffer -> structured array (B, trace_cap)."""
        raw = self.trace.cpu().numpy()
        dt = np.dtype([("p", "<i4"), ("status", "<i4"), ("l", "<i4"), ("kind", "<i4"), ("t1", "<f8"), ("t2", "<f8")])
        return raw.view(dt).reshape(self.B, self.trace_cap)


def shard_range(total, nshards, r):
    """jrlqp_shard_range: (first, count) of shard r of a batch of `total` problems cut into `nshards` contiguous shards."""
    f, c = ct.c_int(0), ct.c_int(0)
    load_library().jrlqp_shard_range(total, nshards, r, ct.byref(f), ct.byref(c))
    return f.value, c.value


class MultiDeviceBatchSolver:
    """jrlqp_gi_solve_batch_multi from one process: the batch is cut into contiguous shards, one per device, each solved on its own
    device with no exchange in between; outputs stay on their devices, `gather_x` assembles x on one."""

    def __init__(self, n, m, use_bounds, B_total, devices, max_iter=500, big_bnd=1e100, streams=None):
        """streams: one torch stream per shard (default: each device's current stream) -- a rehearsal with one device listed twice
        (jrlqp_debug_multi_allow_shared_device) gives each shard a stream of its own."""
        import torch

        self.torch = torch
        self.lib = load_library()
        self.streams = streams
        self.n, self.m, self.nb, self.B = n, m, (n if use_bounds else 0), B_total
        self.devices = [torch.device(d) for d in devices]
        self.ids = (ct.c_int * len(devices))(*[d.index or 0 for d in self.devices])
        self.opts = Options(max_iter, big_bnd)
        self.out = []
        mt = m + self.nb
        for r, dev in enumerate(self.devices):
            _, cnt = shard_range(B_total, len(devices), r)
            wsb = int(self.lib.jrlqp_gi_batch_workspace_bytes_on(self.ids[r], n, m, self.nb, cnt))
            if wsb < 0:
                raise JrlqpError(f"batched path does not support n={n}, m={m}")
            f64, i32 = torch.float64, torch.int32
            self.out.append(dict(x=torch.empty((cnt, n), dtype=f64, device=dev), u=torch.empty((cnt, mt), dtype=f64, device=dev),
                                 status=torch.empty((cnt, mt), dtype=i32, device=dev), term=torch.empty((cnt,), dtype=i32, device=dev),
                                 iters=torch.empty((cnt,), dtype=i32, device=dev), f=torch.empty((cnt,), dtype=f64, device=dev),
                                 ws=torch.empty((wsb + 7) // 8, dtype=f64, device=dev) if wsb else None, wsb=wsb))

    def _shards(self, inputs):
        sh = (ShardBuffers * len(self.devices))()
        for r, (dev, o) in enumerate(zip(self.devices, self.out)):
            i = inputs[r]
            for k in ("G", "a", "C", "bl", "bu", "xl", "xu"):
                setattr(sh[r], "d" + k, i[k].data_ptr() if i.get(k) is not None and i[k].numel() else None)
            sh[r].das = None
            sh[r].dx, sh[r].du, sh[r].dstatus = o["x"].data_ptr(), o["u"].data_ptr() if o["u"].numel() else None, o["status"].data_ptr() if o["status"].numel() else None
            sh[r].dterm, sh[r].diters, sh[r].df, sh[r].dL = o["term"].data_ptr(), o["iters"].data_ptr(), o["f"].data_ptr(), None
            sh[r].dworkspace = o["ws"].data_ptr() if o["ws"] is not None else None
            sh[r].workspace_bytes = o["wsb"]
            sh[r].stream = (self.streams[r] if self.streams else self.torch.cuda.current_stream(dev)).cuda_stream
        return sh

    def solve(self, inputs, synchronize=True):
        """inputs[r]: dict of the shard's tensors on device r (G (cnt, n, n), a, C (cnt, m, n), bl, bu, xl, xu).  Returns self.out."""
        self._sh = self._shards(inputs)
        check(self.lib.jrlqp_gi_solve_batch_multi(0, self.n, self.m, self.nb, self.B, len(self.devices), self.ids, self._sh, ct.byref(self.opts),
                                                  1 if synchronize else 0))
        return self.out

    def gather_x(self, root=0):
        x_all = self.torch.empty((self.B, self.n), dtype=self.torch.float64, device=self.devices[root])
        check(self.lib.jrlqp_gi_gather_x_multi(self.n, self.B, len(self.devices), self.ids, self._sh, root, ct.c_void_p(x_all.data_ptr()), 1))
        return x_all


class DenseSolver:
    """Mode A: one (large) dense QP, state in HBM, host-driven active-set loop (jrlqp_dense_* of the C ABI).
    Mirrors the reference's GoldfarbIdnaniSolver(nbVar, nbCstr, useBounds) + solve(G, a, C, bl, bu, xl, xu)."""

    def __init__(self, n, m, use_bounds, max_iter=500, big_bnd=1e100):
        self.lib = load_library()
        self.n, self.m, self.nb = n, m, (n if use_bounds else 0)
        self.opts = Options(max_iter, big_bnd)
        h = ct.c_void_p()
        check(self.lib.jrlqp_dense_create(n, m, self.nb, ct.byref(h)))
        self.h = h
        self._trace = None

    def set_trace(self, cap):
        """Iteration trace of the next solves (jrlqp_dense_set_trace): `cap` host records, read back with trace_numpy()."""
        self._trace = (TraceRec * cap)() if cap else None
        check(self.lib.jrlqp_dense_set_trace(self.h, ct.cast(self._trace, ct.c_void_p) if cap else None, cap))

    def trace_numpy(self, iters):
        return [(t.p, t.status, t.l, t.kind, t.t1, t.t2) for t in self._trace[: min(iters, len(self._trace))]]

    def set_trace_vectors(self, cap):
        """x, u, z, r, f, q of every pass of the next solves (jrlqp_dense_set_trace_vectors): read back with trace_vectors(iters)."""
        self.lib.jrlqp_dense_trace_vectors_stride.restype = ct.c_int
        self.lib.jrlqp_dense_trace_vectors_stride.argtypes = [ct.c_void_p]
        self.lib.jrlqp_dense_set_trace_vectors.restype = ct.c_int
        self.lib.jrlqp_dense_set_trace_vectors.argtypes = [ct.c_void_p, ct.c_void_p, ct.c_int]
        self._vstride = self.lib.jrlqp_dense_trace_vectors_stride(self.h)
        self._vtrace = np.zeros((max(cap, 1), self._vstride)) if cap else None
        check(self.lib.jrlqp_dense_set_trace_vectors(self.h, self._vtrace.ctypes.data_as(ct.c_void_p) if cap else None, cap))

    def trace_vectors(self, iters):
        """[dict(x, u, z, r, f, q)] per pass; u has q + 1 entries and r has q, by position in the active set."""
        n, mt1 = self.n, self.m + self.nb + 1
        out = []
        for rec in self._vtrace[: min(iters, len(self._vtrace))]:
            q = int(rec[2 * n + 2 * mt1 + 1])
            out.append(dict(x=rec[:n].copy(), u=rec[n : n + q + 1].copy(), z=rec[n + mt1 : 2 * n + mt1].copy(),
                            r=rec[2 * n + mt1 : 2 * n + mt1 + q].copy(), f=float(rec[2 * n + 2 * mt1]), q=q))
        return out

    def sync_mode(self):
        """(mode, reruns): mode 0 = device-side tags, 1 = events; reruns = solves repeated with events after a starved hand-off."""
        r = ct.c_int(0)
        return self.lib.jrlqp_dense_sync_mode(self.h, ct.byref(r)), r.value

    def inject_sync_fault(self, at_pass):
        check(self.lib.jrlqp_debug_dense_inject_sync_fault(self.h, at_pass))

    def solve(self, G, a, C, bl, bu, xl=None, xu=None, warm=False, guess=None):
        """G (n,n) symmetric float64 CUDA tensor -- OVERWRITTEN by its Cholesky factor (lower triangle of the
        column-major matrix, i.e. G.T's lower triangle in torch's row-major view); C (m,n): row i = constraint i."""
        import torch

        n, m = self.n, self.m
        for t in (G, a, C, bl, bu, xl, xu):
            if t is not None:
                assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        term = ct.c_int(-1)
        st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
        common = (self.h, _ptr(G), n, _ptr(a), _ptr(C) if m else None, n, _ptr(bl) if m else None, _ptr(bu) if m else None,
                  _ptr(xl) if self.nb else None, _ptr(xu) if self.nb else None)
        if warm:
            # experimental::GoldfarbIdnaniSolver with warmStart_ (jrlqp_dense_solve_warm): guess = (m + nb) int32 HOST array or None
            g = None if guess is None else np.ascontiguousarray(guess, dtype=np.int32)
            assert g is None or g.shape == (m + self.nb,)
            rc = self.lib.jrlqp_dense_solve_warm(*common, g.ctypes.data_as(ct.c_void_p) if g is not None else None, ct.byref(self.opts),
                                                 ct.byref(term), st)
        else:
            rc = self.lib.jrlqp_dense_solve(*common, ct.byref(self.opts), ct.byref(term), st)
        if rc != 0:
            raise JrlqpError(f"jrlqp_dense_solve failed: {rc} (hip {self.lib.jrlqp_dense_last_hip_error(self.h)})")
        x = np.zeros(n)
        u = np.zeros(m + self.nb)
        status = np.zeros(m + self.nb, dtype=np.int32)
        iters = ct.c_int(0)
        f = ct.c_double(0.0)
        check(self.lib.jrlqp_dense_results(self.h, x.ctypes.data_as(ct.c_void_p), u.ctypes.data_as(ct.c_void_p),
                                           status.ctypes.data_as(ct.c_void_p), ct.byref(iters), ct.byref(f), st))
        return dict(term=term.value, x=x, u=u, status=status, iters=iters.value, f=f.value)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jrlqp_dense_destroy(self.h)
                self.h = None
        except Exception:
            pass


class StructuredBlocks:
    """Plan for the structured Cholesky family over `nprob` identically shaped problems stored as dense (N, N)
    column-major matrices on the device (the usual case in the reference's tests: blocks are views into one big
    matrix, tests/blockArrowLLTTest.cpp:46-47).  kind: "arrow_down", "arrow_up" or "tridiag".

    H: float64 CUDA tensor of shape (nprob, N, N) holding each problem TRANSPOSED w.r.t. torch's row-major view,
    i.e. H[b].T is the matrix (so that the memory is column-major with leading dimension N)."""

    KINDS = {"arrow_down": 0, "arrow_up": 1, "tridiag": 2}

    def __init__(self, kind, block_sizes, H, max_nrhs=8):
        import torch

        self.lib = load_library()
        self.kind = self.KINDS[kind]
        self.bs = [int(b) for b in block_sizes]
        self.nblk = len(self.bs)
        self.N = sum(self.bs)
        assert H.is_cuda and H.dtype == torch.float64 and H.is_contiguous() and H.shape[1:] == (self.N, self.N)
        self.H = H
        self.nprob = H.shape[0]
        self.max_nrhs = max_nrhs
        h = ct.c_void_p()
        arr = (ct.c_int * self.nblk)(*self.bs)
        check(self.lib.jrlqp_blocks_create(self.kind, self.nblk, arr, self.nprob, max_nrhs, ct.byref(h)))
        self.h = h
        off = [0]
        for b in self.bs:
            off.append(off[-1] + b)
        N, base, esz = self.N, H.data_ptr(), 8
        diag, side = [], []
        for p in range(self.nprob):
            pbase = base + p * N * N * esz
            for i in range(self.nblk):
                diag.append(pbase + (off[i] + off[i] * N) * esz)  # element (off_i, off_i), column-major
            for i in range(self.nblk - 1):
                if self.kind == 0:  # last block row: rows off[-2].., columns off[i]..
                    r, c = off[self.nblk - 1], off[i]
                elif self.kind == 1:  # first block column: rows off[i+1].., columns 0..
                    r, c = off[i + 1], 0
                else:  # sub-diagonal: rows off[i+1].., columns off[i]..
                    r, c = off[i + 1], off[i]
                side.append(pbase + (r + c * N) * esz)
        ns = max(self.nblk - 1, 1)
        dptr = (ct.c_void_p * len(diag))(*diag)
        sptr = (ct.c_void_p * max(len(side), 1))(*side) if side else (ct.c_void_p * 1)()
        ldd = (ct.c_int * self.nblk)(*([N] * self.nblk))
        lds = (ct.c_int * ns)(*([N] * ns))
        check(self.lib.jrlqp_blocks_bind(self.h, dptr, ldd, sptr, lds, self._stream()))

    def _stream(self):
        import torch

        return ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def llt(self, fetch=True):
        """In-place factorisation; returns a list of bools (the reference's return value, one per problem); fetch=False: enqueue only."""
        check(self.lib.jrlqp_blocks_llt(self.h, self._stream()))
        if not fetch:
            return None
        ok = (ct.c_int * self.nprob)()
        check(self.lib.jrlqp_blocks_result(self.h, ok, self._stream()))
        return [bool(v) for v in ok]

    def solve(self, M, transpose=False, start=0, end=-1):
        """M: float64 CUDA tensor (nprob, nrhs, N): M[b].T is the N x nrhs right-hand side; solved in place."""
        assert M.is_cuda and M.is_contiguous() and M.shape[0] == self.nprob and M.shape[2] == self.N
        nrhs = M.shape[1]
        check(self.lib.jrlqp_blocks_solve(self.h, int(transpose), _ptr(M), self.N, nrhs * self.N, nrhs, start, end, self._stream()))
        return M

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jrlqp_blocks_destroy(self.h)
                self.h = None
        except Exception:
            pass



class BlockGISolver:
    """Host-side mirror of experimental::BlockGISolver (include/jrl-qp/experimental/BlockGISolver.h:24-40) over a batch of
    `nprob` identically shaped structured QPs.  kind: "tridiag" | "arrow_up" | "arrow_down" (StructuredG::Type);
    block_sizes: diagonal blocks of G; c_shapes: [(n_i, m_i)] of the block-diagonal StructuredC.

    solve(H, a, Cb, bl, bu, xl, xu):
      H   (nprob, N, N) float64 CUDA tensor, H[b].T = G (column-major, leading dimension N); FACTORISED IN PLACE
      Cb  (nprob, ctot) float64 CUDA tensor: the blocks of C back to back, block i stored column-major n_i x m_i
      a (nprob, N); bl, bu (nprob, m); xl, xu (nprob, N) or None.

    form: "auto" | "sequence" | "c9" (include/jrlqp_hip.h, JRLQP_BLOCKGI_FORM_*); `self.form` says what the plan runs (0 sequence, 1 C9:
    the oracle to compare with is orc_blockgi_solve / orc_blockgi_solve_c9)."""

    FORMS = {"auto": -1, "sequence": 0, "c9": 1}

    def __init__(self, kind, block_sizes, c_shapes, use_bounds, nprob, max_iter=500, big_bnd=1e100, device="cuda", form="auto"):
        import torch

        self.lib = load_library()
        self.kind = StructuredBlocks.KINDS[kind]
        self.bs = [int(b) for b in block_sizes]
        self.nblk = len(self.bs)
        self.N = sum(self.bs)
        self._off = [0]
        for b in self.bs:
            self._off.append(self._off[-1] + b)
        self.cshapes = [(int(r), int(c)) for r, c in c_shapes]
        self.m = sum(c for _, c in self.cshapes)
        self.nb = self.N if use_bounds else 0
        self.nprob = nprob
        self.device = torch.device(device)
        self.opts = Options(max_iter, big_bnd)
        h = ct.c_void_p()
        bs = (ct.c_int * self.nblk)(*self.bs)
        ncb = len(self.cshapes)
        cv = (ct.c_int * max(ncb, 1))(*[r for r, _ in self.cshapes])
        cc = (ct.c_int * max(ncb, 1))(*[c for _, c in self.cshapes])
        check(self.lib.jrlqp_blockgi_create_form(self.kind, self.nblk, bs, ncb, cv, cc, int(bool(use_bounds)), nprob, max_iter,
                                                 self.FORMS[form], ct.byref(h)))
        self.h = h
        self.form = self.lib.jrlqp_blockgi_form(self.h)
        self.workspace_bytes = self.lib.jrlqp_blockgi_workspace_bytes(self.h)
        B, N, m, nb = nprob, self.N, self.m, self.nb
        dev = self.device
        self.x = torch.zeros((B, N), dtype=torch.float64, device=dev)
        self.u = torch.zeros((B, max(m + nb, 1)), dtype=torch.float64, device=dev)
        self.status = torch.zeros((B, max(m + nb, 1)), dtype=torch.int32, device=dev)
        self.term = torch.zeros(B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(B, dtype=torch.int32, device=dev)
        self.f = torch.zeros(B, dtype=torch.float64, device=dev)
        self._bound = None

    def _side_shape(self, i):
        """(rows, cols) of off-diagonal block i and its (row, col) origin in the full matrix (block_views of structured_problems.py)."""
        off = self._off
        if self.kind == 0:  # arrow down: G[last block, block i]
            return (self.bs[-1], self.bs[i]), (off[self.nblk - 1], off[i])
        if self.kind == 1:  # arrow up: G[block i + 1, block 0]
            return (self.bs[i + 1], self.bs[0]), (off[i + 1], 0)
        return (self.bs[i + 1], self.bs[i]), (off[i + 1], off[i])  # tridiagonal: G[block i + 1, block i]

    @property
    def packed_len(self):
        return sum(b * b for b in self.bs) + sum(r * c for (r, c), _ in (self._side_shape(i) for i in range(self.nblk - 1)))

    def pack(self, H):
        """The blocks of H (nprob, N, N; H[b].T = G) back to back, each column-major with its own leading dimension: the compact
        storage `solve` also accepts (the reference's StructuredG takes any MatrixRef per block; 1.43 MB instead of 73 MB at c5)."""
        import torch

        parts = []
        off = self._off
        G = H.transpose(1, 2)  # G[b] = H[b].T
        for i in range(self.nblk):
            blk = G[:, off[i] : off[i + 1], off[i] : off[i + 1]]
            parts.append(blk.transpose(1, 2).reshape(self.nprob, -1))  # column-major
        for i in range(self.nblk - 1):
            (r, c), (r0, c0) = self._side_shape(i)
            blk = G[:, r0 : r0 + r, c0 : c0 + c]
            parts.append(blk.transpose(1, 2).reshape(self.nprob, -1))
        return torch.cat(parts, dim=1).contiguous()

    def _bind(self, H, Cb):
        key = (H.data_ptr(), Cb.data_ptr() if Cb is not None else 0)
        if self._bound == key:
            return
        N, esz = self.N, 8
        off = self._off
        packed = H.dim() == 2
        diag, side, cblk = [], [], []
        ctot = sum(r * c for r, c in self.cshapes)
        plen = self.packed_len
        for p in range(self.nprob):
            if packed:
                pbase = H.data_ptr() + p * plen * esz
                o = 0
                for i in range(self.nblk):
                    diag.append(pbase + o * esz)
                    o += self.bs[i] * self.bs[i]
                for i in range(self.nblk - 1):
                    (r, c), _ = self._side_shape(i)
                    side.append(pbase + o * esz)
                    o += r * c
            else:
                pbase = H.data_ptr() + p * N * N * esz
                for i in range(self.nblk):
                    diag.append(pbase + (off[i] + off[i] * N) * esz)
                for i in range(self.nblk - 1):
                    _, (r, c) = self._side_shape(i)
                    side.append(pbase + (r + c * N) * esz)
            o = 0
            for r, c in self.cshapes:
                cblk.append(Cb.data_ptr() + (p * ctot + o) * esz)
                o += r * c
        ns = max(self.nblk - 1, 1)
        ncb = len(self.cshapes)
        dptr = (ct.c_void_p * len(diag))(*diag)
        sptr = (ct.c_void_p * max(len(side), 1))(*side) if side else (ct.c_void_p * 1)()
        cptr = (ct.c_void_p * max(len(cblk), 1))(*cblk) if cblk else (ct.c_void_p * 1)()
        if packed:
            ldd = (ct.c_int * self.nblk)(*self.bs)
            lds = (ct.c_int * ns)(*([self._side_shape(i)[0][0] for i in range(self.nblk - 1)] or [1]))
        else:
            ldd = (ct.c_int * self.nblk)(*([N] * self.nblk))
            lds = (ct.c_int * ns)(*([N] * ns))
        ldc = (ct.c_int * max(ncb, 1))(*[max(r, 1) for r, _ in self.cshapes])
        check(self.lib.jrlqp_blockgi_bind(self.h, dptr, ldd, sptr, lds, cptr, ldc, self._stream()))
        self._bound = key

    def _stream(self):
        import torch

        return ct.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def solve(self, H, a, Cb, bl, bu, xl=None, xu=None, trace_cap=0) -> BatchResult:
        import torch

        assert H.is_cuda and H.dtype == torch.float64 and H.is_contiguous()
        assert H.shape == (self.nprob, self.N, self.N) or H.shape == (self.nprob, self.packed_len)  # full matrices, or pack()'s layout
        self._bind(H, Cb)
        self.trace = None
        if trace_cap:
            self.trace = torch.zeros((self.nprob, trace_cap, ct.sizeof(TraceRec)), dtype=torch.uint8, device=self.device)
        check(self.lib.jrlqp_blockgi_solve(self.h, _ptr(a), _ptr(bl), _ptr(bu), _ptr(xl) if self.nb else None, _ptr(xu) if self.nb else None,
                                           ct.byref(self.opts), _ptr(self.x), _ptr(self.u), _ptr(self.status), _ptr(self.term),
                                           _ptr(self.iters), _ptr(self.f), _ptr(self.trace) if trace_cap else None, trace_cap, self._stream()))
        return BatchResult(self.x, self.u, self.status, self.term, self.iters, self.f, None)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.jrlqp_blockgi_destroy(self.h)
                self.h = None
        except Exception:
            pass



class BoxAndSingleConstraintSolver:
    """Mirror of experimental::BoxAndSingleConstraintSolver (include/jrl-qp/experimental/BoxAndSingleConstraintSolver.h:9-33)
    over a batch:  min 1/2 ||x - x0||^2  s.t.  c^T x >= bl,  xl <= x <= xu  -- jrlqp_box_solve_batch: the closed-form init_ of
    BoxAndSingleConstraintSolver.cpp:46-94 as the kernel's prologue (no factorisation, start at the clamped point with the
    clamped bounds active), then the batched GI loop.  objectiveValue() = 1/2 ||x - x0||^2 like the reference's."""

    def __init__(self, n, B, device="cuda", max_iter=500, big_bnd=1e100):
        import torch

        self.lib = load_library()
        self.n, self.B = n, B
        self.device = torch.device(device)
        self.opts = Options(max_iter, big_bnd)
        f64, i32 = torch.float64, torch.int32
        self.x = torch.empty((B, n), dtype=f64, device=self.device)
        self.u = torch.empty((B, 1 + n), dtype=f64, device=self.device)
        self.status = torch.empty((B, 1 + n), dtype=i32, device=self.device)
        self.term = torch.empty((B,), dtype=i32, device=self.device)
        self.iters = torch.empty((B,), dtype=i32, device=self.device)
        self.f = torch.empty((B,), dtype=f64, device=self.device)
        # beyond 104 variables the large-n kernel keeps R (and past 128 J) in a workspace: allocated HERE, never inside solve()
        wsb = 0
        if n > 64:
            with torch.cuda.device(self.device):
                wsb = int(self.lib.jrlqp_gi_batch_workspace_bytes(n, 1, n, B))
            if wsb < 0:
                raise JrlqpError(f"box solver: n = {n} is not supported", ERR_UNSUPPORTED_SIZE)
        self.workspace_bytes = wsb
        self.workspace = torch.empty((wsb + 7) // 8, dtype=f64, device=self.device) if wsb else None

    def solve(self, x0, c, bl, xl, xu, stream=None):
        """x0, c, xl, xu: (B, n) float64 CUDA tensors; bl: (B,) or (B, 1)."""
        import torch

        x0, c, xl, xu = (t.contiguous() for t in (x0, c, xl, xu))
        bl = bl.reshape(self.B).contiguous()
        for t in (x0, c, xl, xu):
            assert t.is_cuda and t.dtype == torch.float64 and t.shape == (self.B, self.n)
        assert bl.is_cuda and bl.dtype == torch.float64
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(self.lib.jrlqp_box_solve_batch_ws(self.n, self.B, _ptr(x0), _ptr(c), _ptr(bl), _ptr(xl), _ptr(xu), ct.byref(self.opts),
                                                _ptr(self.x), _ptr(self.u), _ptr(self.status), _ptr(self.term), _ptr(self.iters),
                                                _ptr(self.f), None, 0, _ptr(self.workspace), self.workspace_bytes, ct.c_void_p(s.cuda_stream)))
        return BatchResult(self.x, self.u, self.status, self.term, self.iters, self.f, None)


PROFILE_SLOTS, PROFILE_GI_BATCH, PROFILE_BLOCKGI = 32, 0, 1
DENSE_MAX_N = 4096  # JRLQP_DENSE_MAX_N


def set_profile(kernel: int, prof=None):
    """Diagnostic phase profile (jrlqp_debug_set_profile): prof = int64 CUDA tensor (nprob, >= 32) or None to switch off."""
    lib = load_library()
    if prof is None:
        check(lib.jrlqp_debug_set_profile(kernel, None, 0, 0))
    else:
        assert prof.is_cuda and prof.is_contiguous() and prof.dim() == 2 and prof.element_size() == 8
        check(lib.jrlqp_debug_set_profile(kernel, _ptr(prof), prof.shape[1], prof.shape[0]))


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
