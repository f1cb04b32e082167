#!/usr/bin/env python3
"""The verifier's pairing core on the GPU: Miller loop (6.91 B gates) followed by the final exponentiation (3.52 B gates) for a
batch of instances — 10.43 B of the 11.17 B gates of groth16_verify_compressed, all real gates — ciphertexts discarded (device
rate).  The Miller loop's Fq12 output feeds the final exponentiation (outputs read back and staged as inputs: 48 KB per
instance).  Diagnostic tool; correctness of both plans: tools/miller_plan.py and tests/test_gpu_parity.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
t0 = time.time()
miller = gsv.Plan.from_circuit("miller_loop", ["fq12::square_montgomery", "fq12::mul_by_034_montgomery", "pairing::ell_by_constant_montgomery",
                                               "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery"])
fexp = gsv.Plan.from_circuit("final_exp", ["fq12::mul_montgomery", "fq12::square_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::inverse_montgomery"])
print("plans built in %.1f s: miller %d calls / %d gates, final exp %d calls / %d gates" % (
    time.time() - t0, miller.info["n_calls"], miller.info["n_gates"], fexp.info["n_calls"], fexp.info["n_gates"]), flush=True)
total = miller.info["n_gates"] + fexp.info["n_gates"]
for B in [int(x) for x in sys.argv[1:]] or [512]:
    d, f, t, inp = gsv.labels_from_seed(1, miller.info["n_inputs"])
    D, K = np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1))
    # one session at a time: each holds its own staging buffers (two at once do not fit at 256+ instances)
    sm = gsv.Session(eng, miller, B, retain_stream=False)
    for _ in range(2):
        sm.set_garble_inputs(D, K, np.tile(inp, (B, 1, 1)))
        t0 = time.perf_counter()
        sm.garble_streaming(discard=True)
        tm = time.perf_counter() - t0
    f_out = sm.read_outputs()
    sm.close()
    print("B=%d: miller %.2f s (%.3e gates/s)" % (B, tm, B * miller.info["n_gates"] / tm), flush=True)
    sf = gsv.Session(eng, fexp, B, retain_stream=False)
    for _ in range(2):
        sf.set_garble_inputs(D, K, f_out)  # gate ids continue: the final exponentiation follows the Miller loop in the stream
        t0 = time.perf_counter()
        sf.garble_streaming(gate_id_base=miller.info["n_gates"], discard=True)
        tf = time.perf_counter() - t0
    sf.close()
    print("B=%d: miller %.2f s (%.3e gates/s) + final exp %.2f s (%.3e gates/s) = %.2f s -> %.3e gates/s on %d real verifier gates per instance" % (
        B, tm, B * miller.info["n_gates"] / tm, tf, B * fexp.info["n_gates"] / tf, tm + tf, B * total / (tm + tf), total), flush=True)
