import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The C-ABI library and the oracle checkers must exist before any test imports them."""
    import __graft_entry__
    __graft_entry__.build()
    yield


@pytest.fixture
def quiet_logger():
    from rllab_amd.misc import logger
    logger.set_quiet(True)
    yield
    logger.set_quiet(False)
