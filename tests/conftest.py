import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    import garbled_snark_verifier_amd as gsv
    return gsv.Engine(0)  # raises GsvError (GSV_ERR_DEVICE) when there is no HIP device: no CPU fallback
