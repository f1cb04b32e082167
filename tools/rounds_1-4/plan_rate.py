#!/usr/bin/env python3
"""Plan sessions (one launch per component call) against the chain-replay path on the same gates.  Diagnostic tool."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
eng = gsv.Engine(0)
t0 = time.time()
plan = gsv.Plan.from_circuit("fq12_sqmul_chain:%d" % K, ["fq12::mul_montgomery", "fq12::square_montgomery"])
print("plan: %d calls, %d gates, built in %.1f s" % (plan.info["n_calls"], plan.info["n_gates"], time.time() - t0))
n_in = plan.info["n_inputs"]
d, f, t, inp = gsv.labels_from_seed(1, n_in)
D, Kc, I = np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1))
sess = gsv.Session(eng, plan, B)
for _ in range(2):
    sess.set_garble_inputs(D, Kc, I)
    t0 = time.perf_counter()
    sess.garble(0)
    sess.sync()
    dt = time.perf_counter() - t0
print("plan     : %.3f s -> %.3e gates/s (kernel events %.1f ms)" % (dt, B * plan.info["n_gates"] / dt, sess.last_kernel_ms()))
h_plan = sess.ciphertext_hash(0)
out_plan = sess.read_outputs()[0]
sess.close()
prog = gsv.Program.from_circuit("fq12_sqmul", chain_feedback=True)
s2 = gsv.Session(eng, prog, B, K, K)
for _ in range(2):
    s2.set_garble_inputs(D, Kc, I)  # the feedback of a replayed chain overwrites its inputs
    t0 = time.perf_counter()
    s2.garble(0)
    s2.sync()
    dt = time.perf_counter() - t0
print("replayed : %.3f s -> %.3e gates/s" % (dt, B * prog.info["n_gates"] * K / dt))
print("same stream:", h_plan == s2.ciphertext_hash(0), bool((out_plan == s2.read_outputs()[0]).all()))
