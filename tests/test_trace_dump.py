"""MATLAB-syntax iteration dump (jrl-qp_amd/trace.py) in the reference Logger's line format (Logger.h:98-165)."""
import numpy as np


def test_trace_dump_format(pkg, oracle):
    import jrl_qp_amd.trace as T

    G = np.array([[4.0, -2.0], [-2.0, 4.0]])
    r = oracle.solve(G, [6.0, 0.0], np.array([[1.0], [1.0]]), [2.0], [10.0], [0.0, 0.0], [10.0, 10.0], trace_cap=4)
    txt = T.to_matlab(r.trace, name="gi")
    lines = txt.strip().split("\n")
    assert lines[0] == "gi(1).it = 0;" and "gi(1).selectedConstraint = 0;" in lines and "gi(1).status = 1;" in lines
    assert "gi(1).Activate = true;" in lines and "gi(1).l = 0;" in lines
    assert all(ln.endswith(";") and ln.startswith("gi(1).") for ln in lines)  # one iteration (the paper example)
    t2 = [ln for ln in lines if ln.startswith("gi(1).t2")][0]
    assert abs(float(t2.split("=")[1].strip(" ;")) - 5.0) < 1e-12  # full step of length 5 = the multiplier


def test_trace_dump_with_vectors(pkg, oracle):
    """The values of LogFlags::ITERATION_BASIC_DETAILS (x, u, f at the top of a pass; z, r behind the step computation: DualSolver.cpp:108,
    :128-129) from the oracle's vector trace, in the reference's order around the scalar lines."""
    import jrl_qp_amd.trace as T

    G = np.array([[4.0, -2.0], [-2.0, 4.0]])
    r, vec = oracle.solve_with_vectors(G, [6.0, 0.0], np.array([[1.0], [1.0]]), [2.0], [10.0], [0.0, 0.0], [10.0, 10.0], cap=4, trace_cap=4)
    assert len(vec) == r.iters == 1 and vec[0]["q"] == 0 and len(vec[0]["u"]) == 1 and len(vec[0]["r"]) == 0
    assert np.allclose(vec[0]["x"], [-2.0, -1.0])  # the unconstrained minimum the paper's example starts from
    lines = T.to_matlab(r.trace, name="gi", vectors=vec).strip().split("\n")
    names = [ln.split(" = ")[0].split(".")[1] for ln in lines]
    assert names == ["it", "x", "u", "f", "selectedConstraint", "status", "t1", "t2", "z", "r", "t", "Activate", "l"]
    assert lines[1].startswith("gi(1).x = [-2.0, -1.0") and lines[1].endswith("]';")
