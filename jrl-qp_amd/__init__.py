"""jrl-qp_amd -- MI355X-native Goldfarb-Idnani dense QP hot path (drop-in behind jrl-qp's
GoldfarbIdnaniSolver::solve surface) plus the structured block-arrow / block-tridiagonal Cholesky family.

The numerics live in csrc/ (hand-written HIP for gfx950 behind the C ABI of include/jrlqp_hip.h).  This Python
package is plumbing only: it dlopens the C-ABI library, hands it device pointers (torch is used for device
memory, streams and torch.distributed) and provides the synthetic-workload generator and the KKT checker.
There is NO CPU fallback: every entry point raises if the HIP library is missing.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjrlqp_hip.so")

__all__ = ["LIB_PATH", "load_library", "smoke_check"]


def load_library():
    from .capi import load_library as _ll

    return _ll()


def smoke_check(oracle):
    from .capi import smoke_check as _sc

    return _sc(oracle)
