"""QPS reader (Maros-Meszaros convex QP test-set format: MPS + a QUADOBJ / QMATRIX section).

Semantics follow the reference's test-support reader (tests/QPSReader.cpp:172-330 of jrl-umi3218/jrl-qp): free-format
fields, objective row = the first N row, default bounds 0 <= x <= bigBnd, RANGES on E/L/G rows as in the MPS
specification, a RHS entry on the objective row is minus the objective constant, QUADOBJ gives the lower triangle
(symmetrised), QMATRIX the full matrix.  Integer markers and negative-RHS conventions beyond that are not supported
(neither are they in the reference).  Returns the problem in the solver's convention

    min 1/2 x^T G x + a^T x + objCst    s.t.  l <= C x <= u,  xl <= x <= xu          (C: one constraint per ROW here)
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SECTIONS = {"NAME", "ROWS", "COLUMNS", "RHS", "RANGES", "BOUNDS", "QUADOBJ", "QMATRIX", "ENDATA", "OBJSENSE"}


@dataclass
class QPSProblem:
    name: str
    G: np.ndarray
    a: np.ndarray
    C: np.ndarray  # nbCstr x nbVar
    l: np.ndarray
    u: np.ndarray
    xl: np.ndarray
    xu: np.ndarray
    objCst: float
    nbEq: int
    hasFixedVariables: bool
    row_names: list
    col_names: list

    def solver_args(self):
        """(G, a, C (n x m, one constraint per column), bl, bu, xl, xu) as GoldfarbIdnaniSolver::solve takes them."""
        return (np.asfortranarray(self.G), self.a.copy(), np.asfortranarray(self.C.T), self.l.copy(), self.u.copy(), self.xl.copy(), self.xu.copy())


def read_qps(text: str, big_bnd: float = 1e100) -> QPSProblem:
    name, section = "", None
    rows, row_type, obj_row = {}, [], None
    cols = {}
    a_val, c_val, g_val = {}, [], []
    rhs, ranges, bounds = {}, {}, []
    obj_cst, full_obj = 0.0, False
    rhs_name = None
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.rstrip()
        if not line.strip() or line.lstrip().startswith("*"):
            continue
        tok = line.split()
        if not raw[0].isspace() and tok[0].upper() in SECTIONS:
            section = tok[0].upper()
            if section == "NAME":
                name = tok[1] if len(tok) > 1 else ""
            elif section == "QMATRIX":
                full_obj = True
            elif section == "OBJSENSE":
                raise ValueError(f"line {ln}: OBJSENSE is not supported")
            if section == "ENDATA":
                break
            continue
        if section == "ROWS":
            t, rn = tok[0].upper(), tok[1]
            if t == "N":
                if obj_row is None:
                    obj_row = rn
                continue  # further free rows are ignored
            if t not in "ELG":
                raise ValueError(f"line {ln}: unknown row type {t}")
            rows[rn] = len(row_type)
            row_type.append(t)
        elif section == "COLUMNS":
            if len(tok) >= 3 and tok[1] == "'MARKER'":
                raise ValueError(f"line {ln}: integer markers are not supported")
            cn = tok[0]
            j = cols.setdefault(cn, len(cols))
            for rn, v in zip(tok[1::2], tok[2::2]):
                if rn == obj_row:
                    a_val[j] = float(v)
                elif rn in rows:
                    c_val.append((rows[rn], j, float(v)))
                else:
                    raise ValueError(f"line {ln}: unknown row {rn}")
        elif section == "RHS":
            if len(tok) % 2 == 1:  # named RHS vector
                if rhs_name is None:
                    rhs_name = tok[0]
                elif tok[0] != rhs_name:
                    raise ValueError(f"line {ln}: a second RHS vector")
                tok = tok[1:]
            for rn, v in zip(tok[0::2], tok[1::2]):
                if rn == obj_row:
                    obj_cst = -float(v)
                else:
                    rhs[rows[rn]] = float(v)
        elif section == "RANGES":
            if len(tok) % 2 == 1:
                tok = tok[1:]
            for rn, v in zip(tok[0::2], tok[1::2]):
                ranges[rows[rn]] = float(v)
        elif section == "BOUNDS":
            t = tok[0].upper()
            if t in ("FR", "MI", "PL"):
                cn, v = tok[-1], 0.0
            else:
                cn, v = tok[-2], float(tok[-1])
            if t not in ("LO", "UP", "FX", "FR", "MI", "PL"):
                raise ValueError(f"line {ln}: bound type {t} is not supported")
            bounds.append((t, cols[cn], v))
        elif section in ("QUADOBJ", "QMATRIX"):
            g_val.append((cols[tok[0]], cols[tok[1]], float(tok[2])))
        else:
            raise ValueError(f"line {ln}: data outside a section")
    n, m = len(cols), len(row_type)
    G = np.zeros((n, n))
    for i, j, v in g_val:
        G[i, j] = v
        if not full_obj:
            G[j, i] = v
    a = np.zeros(n)
    for j, v in a_val.items():
        a[j] = v
    C = np.zeros((m, n))
    for i, j, v in c_val:
        C[i, j] = v
    l, u = np.zeros(m), np.zeros(m)
    nb_eq = 0
    for i, t in enumerate(row_type):
        b = rhs.get(i, 0.0)
        if t == "E":
            l[i] = u[i] = b
            nb_eq += 1
        elif t == "L":
            l[i], u[i] = -big_bnd, b
        else:
            l[i], u[i] = b, big_bnd
    for i, v in ranges.items():
        t = row_type[i]
        if t == "E":
            if v >= 0:
                u[i] += v
            else:
                l[i] += v
        elif t == "L":
            l[i] = u[i] - abs(v)
        else:
            u[i] = l[i] + abs(v)
    xl, xu = np.zeros(n), np.full(n, big_bnd)
    fixed = False
    for t, j, v in bounds:
        if t == "LO":
            xl[j] = v
        elif t == "UP":
            xu[j] = v
        elif t == "FX":
            xl[j] = xu[j] = v
            fixed = True
        elif t == "FR":
            xl[j], xu[j] = -big_bnd, big_bnd
        elif t == "MI":
            xl[j] = -big_bnd
        elif t == "PL":
            xu[j] = big_bnd
    inv_r = {v: k for k, v in rows.items()}
    inv_c = {v: k for k, v in cols.items()}
    return QPSProblem(name, G, a, C, l, u, xl, xu, obj_cst, nb_eq, fixed, [inv_r[i] for i in range(m)], [inv_c[j] for j in range(n)])
