#!/usr/bin/env python3
"""garble || evaluate on the device, one instance of the verifier (bench.py's `mode_rates.garble_then_evaluate`), repeated: how stable is
the overlap of the two long launches?  Round 4: 39.9 - 65.5 s run to run (the evaluator's stream was probed for a hardware queue of its
own, but the two launches still shared the CUs as the dispatcher saw fit); round 5: CU-masked streams (engine.cpp, ensure_pair).
usage: pair_overlap.py [repeats]     GSV_PAIR_CU_MASK=0 selects the round-4 behaviour."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import garbled_snark_verifier_amd as gsv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
eng = gsv.Engine(0)
path = "/dev/shm/gsv_pair_overlap_%d.gsvplan" % os.getuid()
if not os.path.exists(path):
    t0 = time.time()
    gsv.Plan.build_file(case["circuit"], bench.SMALL_BATCH_UNITS + ["fp254::exp_chunk"], path, window_div=1)
    print("plan built in %.1f s" % (time.time() - t0), flush=True)
plan = gsv.Plan.load(path, eng)
for i in range(n):
    r = bench.garble_then_evaluate(gsv, eng, plan, case, np)
    print("mask=%s run %d: %.2f s, decoded %d, consistent %s, windows %d" % (os.environ.get("GSV_PAIR_CU_MASK", "1"), i, r["seconds"], r["decoded_output"], r["labels_consistent_and_output_label_matches_fixture"], r["windows"]), flush=True)
plan.close()
