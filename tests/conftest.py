import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURES = os.path.join(ROOT, "tests", "golden", "fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure). Built on demand with gcc."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def native():
    """The product C-ABI library (must be built: python -m indexed_bzip2_amd.build)."""
    import indexed_bzip2_amd as m
    from indexed_bzip2_amd import build as b
    b.build()
    m.lib()
    return m


def fixture_names():
    return sorted(f[:-4] for f in os.listdir(FIXTURES) if f.endswith(".bz2"))


def read_fixture(name):
    with open(os.path.join(FIXTURES, name + ".bz2"), "rb") as f:
        enc = f.read()
    with open(os.path.join(FIXTURES, name), "rb") as f:
        raw = f.read()
    return enc, raw
