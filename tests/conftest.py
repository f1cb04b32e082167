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
    config.addinivalue_line("markers", "slow: CPU tests of several minutes each (run with `-m slow`; their log is committed under profiles/)")


def pytest_collection_modifyitems(config, items):
    """`-m slow` runs the slow set; without it (the driver's `-m "not gpu"`) slow tests are skipped unless GSV_SLOW_TESTS=1."""
    import os
    if "slow" in (config.getoption("-m") or "") or os.environ.get("GSV_SLOW_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="slow: run with `-m slow` (or GSV_SLOW_TESTS=1); this round's log: profiles/r04_parity/ref_gadgets_slow.log")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    import garbled_snark_verifier_amd as gsv
    return gsv.Engine(0)  # raises GsvError (GSV_ERR_DEVICE) when there is no HIP device: no CPU fallback
