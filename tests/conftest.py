import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def oracle64():
    """The same oracle built in float64 (oracle/oracle64.py): arbiter for ill-conditioned randomised cases."""
    from oracle import oracle64 as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def oracle_adam():
    from oracle import adam_oracle
    return adam_oracle
