"""Optional cross-check of the oracle's primitives against Eigen itself: built and run only on a box that has
Eigen's headers -- this image does not, and the test then SKIPS (it never fabricates a stand-in).  `make -C oracle eigen` compiles
oracle/eigen_crosscheck.cpp against the real <Eigen/Dense>; the binary compares orc_llt / orc_inv_lt / orc_dot / orc_dot_tree to
tolerance and orc_givens bit for bit with JacobiRotation::makeGivens."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eigen_crosscheck_if_eigen_is_installed():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "eigen"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    exe = os.path.join(ROOT, "oracle", "eigen_crosscheck")
    if "Eigen not found" in out.stdout or not os.path.exists(exe):
        pytest.skip("Eigen is not installed on this box (the reference's only dependency; absent from this image)")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "eigen cross-check OK" in run.stdout, run.stdout + run.stderr
