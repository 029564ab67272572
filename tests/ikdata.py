"""Reader for the reference's MultiIK fixtures (tests/golden/MultiIK.zip, a data file of the reference's
own test-suite: tests/MultiIK.zip, read there by tests/IKmatReader.cpp:95-146)."""
import io
import os
import zipfile

import numpy as np

ZIP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "MultiIK.zip")


def _mat(text):
    return np.loadtxt(io.StringIO(text), ndmin=2)


def read_tri_block_diag():
    """n=387 (9 blocks of 43), m=1621 one-sided constraints C x <= u (BlockGISolverTest.in.cpp:172-188)."""
    z = zipfile.ZipFile(ZIP)
    rd = lambda nm: _mat(z.read("MultiIK/triBlockDiag_%s.txt" % nm).decode())
    G = rd("G")
    C = rd("C").T  # file has one constraint per row; the solver takes n x m
    a = rd("a").ravel()
    u = rd("u").ravel()
    sol = rd("sol").ravel()
    return dict(G=G, a=a, C=np.asfortranarray(C), bl=np.full(len(u), -np.inf), bu=u, sol=sol, block=43, nblk=9)


def read_arrow():
    """n=210 (5 blocks of 42, arrow pointing up), 25 inequalities C x <= d, box bounds (IKmatReader.cpp:106-146)."""
    z = zipfile.ZipFile(ZIP)
    lines = z.read("MultiIK/arrowAllData.txt").decode().splitlines()
    sec = {}
    name = None
    for ln in lines:
        if ln.startswith("===="):
            name = ln.strip("= ").strip()
            sec[name] = []
        else:
            sec[name].append(ln)
    n = int(sec["dim_var"][0])
    mi = int(sec["dim_ineq"][0])
    Q = _mat("\n".join(sec["Q"]))
    c = _mat("\n".join(sec["c"])).ravel()
    Cm = _mat("\n".join(sec["C"]))
    d = _mat("\n".join(sec["d"])).ravel()
    xl = _mat("\n".join(sec["x_min"])).ravel()
    xu = _mat("\n".join(sec["x_max"])).ravel()
    assert Q.shape == (n, n) and Cm.shape == (mi, n)
    return dict(G=Q, a=c, C=np.asfortranarray(Cm.T), bl=np.full(mi, -np.inf), bu=d, xl=xl, xu=xu, block=42, nblk=5)
