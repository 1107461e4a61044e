import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def built():
    """Build the in-tree native code once (HIP library + CPU oracle)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def fiksi(built):
    import fiksi_amd

    return fiksi_amd


@pytest.fixture(scope="session")
def ctx(fiksi):
    """Device context; only GPU tests request it."""
    c = fiksi.Context(0)
    yield c
    c.close()
