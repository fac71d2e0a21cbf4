import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def built(pkg):
    """Libraries built in-tree (hipcc cross-compiles without a GPU)."""
    so = os.path.join(entry.PKG_DIR, "lib", "libcln_amd.so")
    if not os.path.exists(so) or os.environ.get("CLN_AMD_REBUILD") == "1":
        pkg.build()
    return pkg


@pytest.fixture(scope="session")
def oracle():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
