import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import frw_testlib
    return frw_testlib.load_oracle()


@pytest.fixture(scope="session")
def engine():
    """The product engine on device 0.  No fallback: a missing library or GPU fails the test."""
    import falcon_r1cs_amd
    eng = falcon_r1cs_amd.WitnessEngine(0)
    yield eng
    eng.close()
