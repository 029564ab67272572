"""QPS reader (jrl-qp_amd/qps.py; semantics of the reference's tests/QPSReader.cpp:172-330) on the example of the
Maros-Meszaros QPS format description, solved by the oracle (CPU) and by the batched GPU kernel.  The Maros-Meszaros data
files themselves are not part of the reference repository (tests need WITH_QPS_TESTS + QPS_DIR, tests/CMakeLists.txt:7-15)."""
import numpy as np
import pytest

QPTEST = """\
NAME          QPTEST
ROWS
 N  obj
 G  r1
 L  r2
COLUMNS
    c1        r1                 2.0   r2                -1.0
    c1        obj                1.5
    c2        r1                 1.0   r2                 2.0
    c2        obj               -2.0
RHS
    rhs1      obj               -4.0
    rhs1      r1                 2.0   r2                 6.0
BOUNDS
 UP bnd1      c1                20.0
QUADOBJ
    c1        c1                 8.0
    c1        c2                 2.0
    c2        c2                10.0
ENDATA
"""


@pytest.fixture(scope="module")
def Q(pkg):
    import jrl_qp_amd.qps as qps

    return qps


def test_parse_example(Q):
    p = Q.read_qps(QPTEST)
    assert p.name == "QPTEST" and p.row_names == ["r1", "r2"] and p.col_names == ["c1", "c2"]
    assert np.array_equal(p.G, [[8.0, 2.0], [2.0, 10.0]]) and np.array_equal(p.a, [1.5, -2.0]) and p.objCst == 4.0
    assert np.array_equal(p.C, [[2.0, 1.0], [-1.0, 2.0]])
    assert list(p.l) == [2.0, -1e100] and list(p.u) == [1e100, 6.0]
    assert list(p.xl) == [0.0, 0.0] and list(p.xu) == [20.0, 1e100]
    assert p.nbEq == 0 and not p.hasFixedVariables


def test_ranges_bounds_and_sections(Q):
    txt = QPTEST.replace("BOUNDS\n", "RANGES\n    rng       r1                 3.0   r2                -4.0\nBOUNDS\n FX bnd1      c2                 0.5\n MI bnd1      c1\n")
    p = Q.read_qps(txt)
    assert list(p.l) == [2.0, 2.0] and list(p.u) == [5.0, 6.0]  # G row: u = l + |R|; L row: l = u - |R|
    assert p.xl[1] == p.xu[1] == 0.5 and p.hasFixedVariables and p.xl[0] == -1e100 and p.xu[0] == 20.0
    with pytest.raises(ValueError):
        Q.read_qps(QPTEST.replace(" G  r1", " X  r1"))


def test_example_solved_by_the_oracle(Q, oracle):
    """Published optimum of the format example: x = (0.7625, 0.475), objective 8.37187500."""
    p = Q.read_qps(QPTEST)
    G, a, C, bl, bu, xl, xu = p.solver_args()
    r = oracle.solve(G, a, C, bl, bu, xl, xu)
    assert r.term == 0
    assert np.allclose(r.x, [0.7625, 0.475], rtol=0, atol=1e-12)
    assert abs(r.f + p.objCst - 8.371875) < 1e-12


@pytest.mark.gpu
def test_example_solved_on_the_gpu(Q, oracle, pkg):
    import torch

    import jrl_qp_amd.capi as capi

    p = Q.read_qps(QPTEST)
    G, a, C, bl, bu, xl, xu = p.solver_args()
    dev = lambda v: torch.tensor(np.ascontiguousarray(v)[None], device="cuda")
    s = capi.BatchSolver(2, 2, True, 1)
    r = s.solve(dev(G.T), dev(a), dev(C.T), dev(bl), dev(bu), dev(xl), dev(xu))
    torch.cuda.synchronize()
    ref = oracle.solve(G, a, C, bl, bu, xl, xu)
    assert int(r.term[0]) == 0 and np.array_equal(r.x[0].cpu().numpy(), ref.x) and float(r.f[0]) == ref.f
