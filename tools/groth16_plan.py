#!/usr/bin/env python3
"""The WHOLE Groth16 verifier circuit (groth16_verify, src/gadgets/groth16.rs:58-110: window-10 MSM over constant bases,
projective -> affine, Miller loop, final exponentiation, comparison with the constant alpha-beta) on the GPU as one plan.

1. garble two instances with the stream drained and hashed call by call: CBC-MAC, output label and counts == the fixture the
   CPU oracle produced from the FLAT stream (tests/golden/groth16_verify_golden.json, tests/golden/make_big_golden.py);
2. garble two instances with the stream retained in HBM (48 GB each) and EVALUATE them: the valid proof decodes to 1, the same
   proof with one public-input bit flipped to 0;
3. device rate with the ciphertexts discarded.
`--compressed`: the same for groth16_verify_compressed (point decompression in front: the reference's headline circuit).
The verifying key / proof are the synthetic instance of tests/groth16_ref.py; the circuit name in the fixture carries the key."""
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

UNITS = ["fq12::square_montgomery", "fq12::mul_montgomery", "fq12::cyclotomic_square_montgomery", "fq12::mul_by_034_montgomery",
                  "pairing::ell_by_constant_montgomery", "pairing::double_in_place_circuit_montgomery", "pairing::add_in_place_montgomery", "pairing::mul_by_char_montgomery",
                  "bigint::multiplexer", "g1::add_montgomery",
                  # the Fq inversions (binary extended Euclid, fp254impl.rs:333-690) enter as their own 4-iteration components: as ONE unit an
                  # inversion (11 M ciphertexts) — or the Fq12 inversion around it (21 M) — would set the size of every instance's device
                  # ciphertext block (340 MB x 512 instances); their chunks keep the largest block at an Fq12 multiplication's 5.4 M records
                  "inverse::iteration_group", "inverse::divide_chains"]
COMPRESSED = "--compressed" in sys.argv  # groth16_verify_compressed (groth16.rs:250-268): decompression of A, B, C in front of the verifier
if COMPRESSED:  # the square roots are ladders of ~380 Fq multiplications each: four ladder steps (one fp254::exp_chunk component) are a unit
    UNITS += ["fp254::exp_chunk"]
case = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_verify_compressed_golden.json" if COMPRESSED else "groth16_verify_golden.json")))
eng = gsv.Engine(0)
t0 = time.time()
plan = gsv.Plan.from_circuit(case["circuit"], UNITS)
print("plan: %d calls, %d gates, %d ciphertexts, built in %.1f s, host peak RSS %.1f GB" % (
    plan.info["n_calls"], plan.info["n_gates"], plan.info["n_ciphertexts"], time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6), flush=True)
n_in = plan.info["n_inputs"]
assert n_in == case["n_inputs"]
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
print("garble + drain of 2 instances: %.1f s; hash / output label / counts == oracle fixture: %s" % (time.time() - t0, ok), flush=True)
sess.close()

if "--no-eval" not in sys.argv:
    bits_ok = np.unpackbits(np.frombuffer(bytes.fromhex(case["input_bits_hex"]), np.uint8), bitorder="little")[:n_in].astype(np.uint8)
    bits_bad = bits_ok.copy(); bits_bad[case.get("tamper_bit", 0)] ^= 1  # another public input (compressed: A's other root): the proof no longer verifies
    bits = np.stack([bits_ok, bits_bad])
    sess = gsv.Session(eng, plan, 2)
    sess.set_garble_inputs(delta, consts, inputs)
    t0 = time.time()
    sess.garble(0); sess.sync()
    out0 = sess.read_outputs()
    active = np.where(bits[:, :, None] == 1, inputs ^ delta[:, None, :], inputs)
    sess.set_evaluate_inputs(np.stack([consts[:, 0], consts[:, 1] ^ delta], axis=1), active, bits)
    sess.evaluate(0); sess.sync()
    oa, ob = sess.read_outputs(with_bits=True)
    labels_ok = bool((oa == np.where(ob[:, :, None] == 1, out0 ^ delta[:, None, :], out0)).all())
    print("garble + evaluate with the stream in HBM: %.1f s; verifier output (valid proof, tampered input) = (%d, %d), expected (%d, 0); active output labels consistent: %s" % (
        time.time() - t0, ob[0][0], ob[1][0], case["expected_output"], labels_ok), flush=True)
    sess.close()

for B in [int(x) for x in sys.argv[1:] if x.isdigit()] or [256]:
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    sess = gsv.Session(eng, plan, B, retain_stream=False)
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    for _ in range(2):
        t0 = time.perf_counter()
        sess.garble_streaming(discard=True)
        dt = time.perf_counter() - t0
    print("B=%d: %.2f s -> %.3e gates/s on the whole verifier (%d gates per instance)" % (B, dt, B * plan.info["n_gates"] / dt, plan.info["n_gates"]), flush=True)
    sess.close()
