#!/usr/bin/env python3
"""The whole verifier drained through the commitment (garble_streaming, hashes only) with and without the ciphertext ring
(GSV_CT_RING, read when a session is created; RING_AB="ring:priority,..." also toggles
GSV_SIDE_STREAM_PRIORITY): 1 and 16 instances of the small-batch plan, seeds of the cc16 fixture, every
commitment compared with the fixture.  ring_ab.py [instances ...]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import garbled_snark_verifier_amd as gsv

case = json.load(open(os.path.join(ROOT, "tests", "golden", bench.FIXTURE["verifier_compressed"])))
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cc16_verifier_golden.json")))
eng = gsv.Engine(0)
d = tempfile.mkdtemp(prefix="gsv_plan_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "v.gsvplan")
gsv.Plan.build_file(case["circuit"], bench.SMALL_BATCH_UNITS + ["fp254::exp_chunk"], path, window_div=4)
plan = gsv.Plan.load(path, eng)
os.remove(path)
os.environ["GSV_DRAIN_STATS"] = "1"
n_in, gates = plan.info["n_inputs"], plan.info["n_gates"]
for B in [int(x) for x in sys.argv[1:]] or [1, 16]:
    labs = [gsv.labels_from_seed(int(s), n_in) for s in gold["seeds"][:B]]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    for ring, prio in [x.split(":") for x in os.environ.get("RING_AB", "1:1,0:1,1:1,0:1").split(",")]:
        os.environ["GSV_CT_RING"], os.environ["GSV_SIDE_STREAM_PRIORITY"] = ring, prio
        t0 = time.perf_counter()
        sess = gsv.Session(eng, plan, B, retain_stream=False)
        t1 = time.perf_counter()
        sess.set_garble_inputs(delta, consts, inputs)
        hashes = sess.garble_streaming()
        t2 = time.perf_counter()
        si = sess.schedule_info()
        ok = all(hashes[i].hex() == gold["ct_hashes"][i] for i in range(B))
        sess.close()
        print("B=%2d ring=%s side-stream priority=%s: %d windows, %d segments, ring %d records; session %.2f s, pass %.2f s -> %.3e gates/s; commitments %s" % (
            B, ring, prio, si["n_windows"], si["n_segments"], si["ct_ring_records"], t1 - t0, t2 - t1, B * gates / (t2 - t1), "== fixture" if ok else "DIFFER"), flush=True)
