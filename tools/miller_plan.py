#!/usr/bin/env python3
"""The Groth16 Miller loop (6,909,061,143 gates; constant Q = +-G2 generator for two pairs, wire Q for the third) as a plan of
component programs: build, garble two instances with the stream drained + hashed and compare with the oracle's fixture
(tests/golden/miller_loop_golden.json), then the device rate with the ciphertexts discarded.  Diagnostic tool: the 178 line
functions with constant coefficients are 178 different programs (constants are baked into the shift-add schedules)."""
import hashlib
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

UNITS = ["fq12::square_montgomery", "fq12::mul_by_034_montgomery", "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery",
         "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery"]
eng = gsv.Engine(0)
t0 = time.time()
plan = gsv.Plan.from_circuit("miller_loop", UNITS)
print("plan: %d calls, %d gates, %d ciphertexts, built in %.1f s, host peak RSS %.1f GB" % (
    plan.info["n_calls"], plan.info["n_gates"], plan.info["n_ciphertexts"], time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
n_in = plan.info["n_inputs"]
gpath = os.path.join(ROOT, "tests", "golden", "miller_loop_golden.json")
if os.path.exists(gpath):
    case = json.load(open(gpath))
    seeds = [case["seed"], case["seed"] + 1]
    labs = [gsv.labels_from_seed(s, n_in) for s in seeds]
    delta = np.stack([x[0] for x in labs]); consts = np.stack([np.stack([x[1], x[2]]) for x in labs]); inputs = np.stack([x[3] for x in labs])
    sess = gsv.Session(eng, plan, 2, retain_stream=False)
    sess.set_garble_inputs(delta, consts, inputs)
    t0 = time.time()
    hashes = sess.garble_streaming(threads=2)
    out = sess.read_outputs()
    ok = (hashes[0].hex() == case["ct_hash"] and hashlib.sha256(out[0].tobytes()).hexdigest() == case["output_label0_sha256"] and plan.info["n_gates"] == case["gates"]
          and plan.info["n_ciphertexts"] == case["n_ciphertexts"])
    print("garble + drain of 2 instances: %.1f s; hash / output labels / counts == oracle fixture: %s" % (time.time() - t0, ok), flush=True)
    sess.close()
for B in [int(x) for x in sys.argv[1:]] or [256]:
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    sess = gsv.Session(eng, plan, B, retain_stream=False)
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    for _ in range(2):
        t0 = time.perf_counter()
        sess.garble_streaming(discard=True)
        dt = time.perf_counter() - t0
    print("B=%d: %.2f s -> %.3e gates/s" % (B, dt, B * plan.info["n_gates"] / dt), flush=True)
    sess.close()
