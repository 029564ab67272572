import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit (pytest-timeout where it is installed; a no-op elsewhere)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory jrl-qp_amd/, imported as jrl_qp_amd)."""
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import get_oracle

    return get_oracle()
