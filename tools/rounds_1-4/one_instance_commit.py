#!/usr/bin/env python3
"""ONE instance of the verifier with its whole stream folded into one CBC-MAC chain (BASELINE's single-instance figure is stated with the
ciphertext hash): how the size of the ciphertext window decides how much of the serial chain (27 s) hides behind the garbling (31 s).
usage: one_instance_commit.py [window_ct_records ...]   (0 = the session's default: as few windows as the memory allows)"""
import hashlib
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
print("plan in %.1f s" % (time.time() - t0), flush=True)
gates = plan.info["n_gates"]
for wct in [int(x) for x in sys.argv[1:]] or [1 << 26, 0]:
    w = bench.VerifierWork(gsv, eng, plan, 1, [case["seed"]], window_ct_records=wct)
    dt = w.run_pass(commit=True)
    ok = w.ct_hashes[0].hex() == case["ct_hash"] and hashlib.sha256(w.sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
    print("window_ct_records %10d: %d windows, %.2f s -> %.3e gates/s (x%.1f the published 3.2e7), hash + output label == fixture: %s" % (
        wct, w.sess.schedule_info()["n_windows"], dt, gates / dt, gates / dt / 32e6, ok), flush=True)
    w.close()
