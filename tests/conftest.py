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
    config.addinivalue_line("markers", "slow: CPU tests of several minutes each (run with `-m slow`; their log is committed under profiles/); with `gpu`: sub-circuits of the "
                            "verifier as plans of their own (the whole verifier is in the default GPU set; `-m 'gpu and slow'` runs them)")


def pytest_collection_modifyitems(config, items):
    """`-m slow` runs the slow set; without it (the driver's `-m "not gpu"`) slow tests are skipped unless GSV_SLOW_TESTS=1."""
    import os
    if "slow" in (config.getoption("-m") or "") or os.environ.get("GSV_SLOW_TESTS") == "1":
        return
    skip = pytest.mark.skip(reason="slow: run with `-m slow` / `-m 'gpu and slow'` (or GSV_SLOW_TESTS=1); the last round's log: profiles/r06_parity/slow_set.log")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    import garbled_snark_verifier_amd as gsv
    return gsv.Engine(0)  # raises GsvError (GSV_ERR_DEVICE) when there is no HIP device: no CPU fallback


# The headline circuit's units (bench.py VERIFIER_UNITS + the decompression ladders' chunks): ONE plan file of it per test session.
VERIFIER_PLAN_UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery",
                       "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                       "bigint::multiplexer", "g1::add_montgomery", "inverse::iteration_group", "inverse::divide_chains", "fp254::exp_chunk"]


@pytest.fixture(scope="session")
def verifier_plan_file():
    """The built-in builder's plan file (gsv_plan_build_file, window_div 4) of groth16_verify_compressed with one public input, built ONCE per
    session the way bench.py gets it on a fresh machine (~55 s, ~17 GB of host memory, 41.8 GB of records): tests/test_ext_host.py
    compares the plan an external host records through the C ABI with it, tests/test_gpu_parity.py loads it into the GPU."""
    import json
    import shutil
    import tempfile
    import time
    import garbled_snark_verifier_amd as gsv
    case = json.load(open(os.path.join(HERE, "golden", "groth16_verify_compressed_1pub_golden.json")))
    shm = os.path.isdir("/dev/shm") and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > 100e9
    d = tempfile.mkdtemp(prefix="gsv_test_plan_", dir="/dev/shm" if shm else None)
    try:
        path = os.path.join(d, "verifier.gsvplan")
        t0 = time.time()
        gsv.Plan.build_file(case["circuit"], VERIFIER_PLAN_UNITS, path, window_div=4)
        yield {"path": path, "build_s": time.time() - t0, "case": case}
    finally:
        shutil.rmtree(d, ignore_errors=True)
