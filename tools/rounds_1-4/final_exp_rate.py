#!/usr/bin/env python3
"""Device rate on a REAL verifier component: final_exponentiation_montgomery (3,519,328,217 gates) as a 286-call plan,
ciphertexts discarded (the HBM-resident rate; with the drain the run is PCIe-bound, see tools/rounds_1-4/e2e_streaming.py).  Diagnostic tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
t0 = time.time()
plan = gsv.Plan.from_circuit("final_exp", ["fq12::mul_montgomery", "fq12::square_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::inverse_montgomery"])
print("plan: %d calls, %d gates, %d ciphertexts, built in %.1f s" % (plan.info["n_calls"], plan.info["n_gates"], plan.info["n_ciphertexts"], time.time() - t0))
n_in = plan.info["n_inputs"]
for B in [int(x) for x in sys.argv[1:]] or [256, 512]:
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    sess = gsv.Session(eng, plan, B, retain_stream=False)
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    for _ in range(2):
        t0 = time.perf_counter()
        sess.garble_streaming(discard=True)
        dt = time.perf_counter() - t0
    print("B=%d: %.2f s -> %.3e gates/s" % (B, dt, B * plan.info["n_gates"] / dt))
    sess.close()
