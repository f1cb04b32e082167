#!/usr/bin/env python3
"""ONE instance of the verifier, ciphertexts into HBM (no drain): device time of a whole pass as a function of the window size of the
session's schedule — every window boundary ends the overlap of the instance's independent call chains.  The plan file is built by the
library under test (GSV_ENGINE_SO), so several builds can be compared on one box: one_instance_windows.py [window_ct_records ...]
(0 = the session's default)."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import garbled_snark_verifier_amd as gsv

case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
eng = gsv.Engine(0)
d = tempfile.mkdtemp(prefix="gsv_plan_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "v.gsvplan")
t0 = time.time()
gsv.Plan.build_file(case["circuit"], bench.VERIFIER_UNITS + ["fp254::exp_chunk"], path, window_div=4)
plan = gsv.Plan.load(path, eng)
os.remove(path)
print("library %s: plan in %.1f s" % (os.environ.get("GSV_ENGINE_SO", "libgsv_engine.so"), time.time() - t0), flush=True)
gates = plan.info["n_gates"]
for B in [int(x) for x in os.environ.get("OW_INSTANCES", "1").split(",")]:
    for wct in [int(x) for x in sys.argv[1:]] or [0, 1 << 26, 1 << 31]:
        w = bench.VerifierWork(gsv, eng, plan, B, [case["seed"]] + bench.instance_seeds(0, B)[1:], window_ct_records=wct)
        best = min(w.run_pass() for _ in range(2))
        si = w.sess.schedule_info()
        print("B=%d window_ct_records %10d: %3d windows, width %d, depth %d steps: %.2f s -> %.3e gates/s" % (B, wct, si["n_windows"], si["max_width"], si["critical_steps"], best, B * gates / best), flush=True)
        w.close()
