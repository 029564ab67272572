"""The host C++17 mirror of the reference interface (include/jrlqp/GoldfarbIdnaniSolver.hpp): it compiles against
the C ABI on a CPU-only box, and on the GPU box the reference's paper test written against it passes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_host_class.cpp")
SRC_BLOCK = os.path.join(ROOT, "tests", "cpp", "test_block_host_class.cpp")
LIBDIR = os.path.join(ROOT, "jrl-qp_amd", "csrc")


def _build(out, src=SRC):
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", src,
           "-o", out, "-L", LIBDIR, "-ljrlqp_hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_host_class_compiles_and_links(pkg, tmp_path):
    import jrl_qp_amd.capi as capi

    capi.load_library()
    _build(str(tmp_path / "host_class"))


@pytest.mark.gpu
def test_host_class_paper_example(pkg, tmp_path):
    exe = str(tmp_path / "host_class")
    _build(exe)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host class OK" in out.stdout, out.stdout + out.stderr


def test_block_host_class_compiles_and_links(pkg, tmp_path):
    import jrl_qp_amd.capi as capi

    capi.load_library()
    _build(str(tmp_path / "block_host_class"), SRC_BLOCK)


@pytest.mark.gpu
def test_block_host_class_reference_tests(pkg, tmp_path):
    """tests/BlockGISolverTest.in.cpp:70-170 through include/jrlqp/BlockGISolver.hpp: structured == dense (1e-8)."""
    exe = str(tmp_path / "block_host_class")
    _build(exe, SRC_BLOCK)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "block host class OK" in out.stdout, out.stdout + out.stderr
