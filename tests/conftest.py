import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bait_text():
    from tests.util_data import make_bait
    return make_bait()


@pytest.fixture(scope="session")
def built_lib():
    """Make sure the product library and CLI exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from mitoflex_amd import mitofilter
    return mitofilter.load()
