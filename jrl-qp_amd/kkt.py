"""KKT acceptance test, restating the reference's checker (src/test/kkt.cpp:14-22, :87-186;
defaults include/jrl-qp/test/kkt.h:82-84: tau_p = tau_d = 1e-6).

Sign convention: multipliers are <= 0 at lower bounds, >= 0 at upper bounds and
G x + a + C u_c + u_b = 0.  C is n x m (one constraint per column) as the solver takes it.
"""
from __future__ import annotations

import numpy as np


def kkt_residuals(x, u, G, a, C, bl, bu, xl=None, xu=None):
    """Return (stationarity inf-norm, max primal violation, max complementarity violation)."""
    x = np.asarray(x, float)
    u = np.asarray(u, float)
    m = len(bl)
    nb = 0 if xl is None else len(xl)
    dL = G @ x + a + C @ u[:m]
    if nb:
        dL = dL + u[m:]
    cx = C.T @ x
    lo = np.concatenate([bl, xl]) if nb else np.asarray(bl)
    hi = np.concatenate([bu, xu]) if nb else np.asarray(bu)
    val = np.concatenate([cx, x]) if nb else cx
    with np.errstate(invalid="ignore"):
        viol = np.maximum(np.maximum(lo - val, val - hi), 0.0)
        gap_lo = np.where(np.isfinite(lo), np.abs(val - lo), np.inf)
        gap_hi = np.where(np.isfinite(hi), np.abs(val - hi), np.inf)
    compl = np.where(u < 0, gap_lo, np.where(u > 0, gap_hi, 0.0)) * (np.abs(u) > 0)
    return (
        float(np.max(np.abs(dL))) if len(dL) else 0.0,
        float(np.max(viol)) if len(viol) else 0.0,
        float(np.max(compl)) if len(compl) else 0.0,
    )


def _check_constraint(cx, bl, bu, u, tau_x, tau_u):
    li = cx - bl
    ui = cx - bu
    b1 = abs(li) <= tau_x and u <= -tau_u
    b2 = li >= -tau_x and ui <= tau_x and abs(u) <= tau_u
    b3 = abs(ui) <= tau_x and u >= tau_u
    return b1 or b2 or b3


def test_kkt(x, u, G, a, C, bl, bu, xl=None, xu=None, tau_p=1e-6, tau_d=1e-6) -> bool:
    """testKKT (kkt.cpp:87-104): stationarity (:111-142) and per-constraint feasibility/complementarity (:152-186)."""
    x = np.asarray(x, float)
    u = np.asarray(u, float)
    m = len(bl)
    nb = 0 if xl is None else len(xl)
    unorm = float(np.max(np.abs(u))) if len(u) else 0.0
    xnorm = float(np.max(np.abs(x))) if len(x) else 0.0
    tau_u = tau_d * (1 + unorm)
    tau_x = tau_p * (1 + xnorm)
    dL = G @ x + a + C @ u[:m]
    if nb:
        dL = dL + u[m:]
    if len(dL) and float(np.max(np.abs(dL))) > tau_u:
        return False
    cx = C.T @ x
    for i in range(m):
        if not _check_constraint(cx[i], bl[i], bu[i], u[i], tau_x, tau_u):
            return False
    for i in range(nb):
        if not _check_constraint(x[i], xl[i], xu[i], u[m + i], tau_x, tau_u):
            return False
    return True


test_kkt.__test__ = False  # not a pytest test
