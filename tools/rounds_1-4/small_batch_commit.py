#!/usr/bin/env python3
"""BASELINE configs 4 and 5 WITH the ciphertext commitment on one GPU, and what the drain pipeline does for them: one instance (one serial
CBC-MAC chain of 2.98e9 blocks beside 31 s of garbling) and sixteen (cut_and_choose_commit on the full verifier, all 16 records against the
oracle-built fixture), each with the old drain (one gate-order buffer; sixteen instances hashed four chains to a worker) and the
round-4 one (several buffers: the host side may lag the device by up to eight windows; one chain per worker while there is a core per
instance).  GSV_DRAIN_STATS=1 prints where the pipeline waited."""
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
from garbled_snark_verifier_amd import sharding

case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cc16_verifier_golden.json")))
eng = gsv.Engine(0)
d = tempfile.mkdtemp(prefix="gsv_plan_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "v.gsvplan")
t0 = time.time()
gsv.Plan.build_file(case["circuit"], bench.VERIFIER_UNITS + ["fp254::exp_chunk"], path, window_div=4)
plan = gsv.Plan.load(path, eng)
os.remove(path)
print("plan in %.1f s" % (time.time() - t0), flush=True)
gates = plan.info["n_gates"]
os.environ["GSV_DRAIN_STATS"] = "1"
w = bench.VerifierWork(gsv, eng, plan, 1, [case["seed"]])
dt = w.run_pass()
print("1 instance, ciphertexts into HBM: %.2f s -> %.3e gates/s" % (dt, gates / dt), flush=True)
w.close()
for label, env in (("old drain (1 buffer)", {"GSV_DRAIN_DEPTH": "1"}), ("pipeline (default)", {})):
    for k, v in env.items():
        os.environ[k] = v
    w = bench.VerifierWork(gsv, eng, plan, 1, [case["seed"]])
    dt = w.run_pass(commit=True)
    ok = w.ct_hashes[0].hex() == case["ct_hash"] and hashlib.sha256(w.sess.read_outputs()[0].tobytes()).hexdigest() == case["output_label0_sha256"]
    print("1 instance with the commitment, %s: %d windows, %.2f s -> %.3e gates/s (x%.1f the published 3.2e7), hash + output label == fixture: %s" % (
        label, w.sess.schedule_info()["n_windows"], dt, gates / dt, gates / dt / 32e6, ok), flush=True)
    w.close()
    for k in env:
        os.environ.pop(k)
for label, env in (("old drain (1 buffer, four chains per worker)", {"GSV_DRAIN_DEPTH": "1", "GSV_DRAIN_GROUP": "4"}), ("pipeline (default)", {})):
    for k, v in env.items():
        os.environ[k] = v
    t0 = time.perf_counter()
    table, seeds = sharding.cut_and_choose_commit(case["circuit"], gold["master_seed"], gold["total"], 0, 1, engine=eng, program=plan)
    dt = time.perf_counter() - t0
    ok = [hashlib.sha256(r.tobytes()).hexdigest() for r in table] == gold["record_sha256"]
    print("16 instances (cc16) with the commitments, %s: %.2f s -> %.3e gates/s, all 16 records == oracle fixture: %s" % (label, dt, 16 * gates / dt, ok), flush=True)
    for k in env:
        os.environ.pop(k)
