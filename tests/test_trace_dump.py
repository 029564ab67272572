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
