#!/usr/bin/env python3
"""Where the seconds of `sharding.garble_and_commit` go for 16 instances of the verifier besides the drain itself (session creation =
device allocations, label derivation, records, teardown).  Diagnostic."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import garbled_snark_verifier_amd as gsv
from garbled_snark_verifier_amd import sharding

case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cc16_verifier_golden.json")))
eng = gsv.Engine(0)
d = tempfile.mkdtemp(prefix="gsv_plan_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "v.gsvplan")
gsv.Plan.build_file(case["circuit"], bench.VERIFIER_UNITS + ["fp254::exp_chunk"], path, window_div=4)
plan = gsv.Plan.load(path, eng)
os.remove(path)
os.environ["GSV_DRAIN_STATS"] = "1"
B, n_in = 16, plan.info["n_inputs"]
for rep in range(2):
    t = [time.perf_counter()]
    labs = [gsv.labels_from_seed(int(s), n_in) for s in gold["seeds"]]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    t.append(time.perf_counter())
    sess = gsv.Session(eng, plan, B, retain_stream=False)
    t.append(time.perf_counter())
    sess.set_garble_inputs(delta, consts, inputs)
    t.append(time.perf_counter())
    hashes = sess.garble_streaming()
    t.append(time.perf_counter())
    outs = sess.read_outputs()
    recs = np.stack([sharding.commit_record(i, hashes[i], outs[i], delta[i], consts[i, 0], consts[i, 1], inputs[i]) for i in range(B)])
    t.append(time.perf_counter())
    sess.close()
    t.append(time.perf_counter())
    names = ["labels from seeds", "session create", "set inputs", "garble_streaming (incl. buffer allocation)", "outputs + records", "session close"]
    print("pass %d: " % rep + "; ".join("%s %.2f s" % (n, t[i + 1] - t[i]) for i, n in enumerate(names)) + "; total %.2f s; %s" % (t[-1] - t[0], sess and ""), flush=True)
