"""Generates the fixtures of the multi-billion-gate circuits from the CPU oracle's FLAT stream (minutes each):

  python tests/golden/make_big_golden.py miller_loop       -> miller_loop_golden.json      (6.9 B gates, ~4 min)
  python tests/golden/make_big_golden.py groth16_verify    -> groth16_verify_golden.json   (the whole verifier, ~7 min)
  python tests/golden/make_big_golden.py groth16_verify_compressed -> groth16_verify_compressed_golden.json   (with point decompression)
  python tests/golden/make_big_golden.py groth16_verify_compressed_1pub -> groth16_verify_compressed_1pub_golden.json
        (ONE public input = the configuration the reference quotes its gate count and timings on; bench.py's default)

The Groth16 instance (verifying key, proof, public inputs) is the synthetic one of tests/groth16_ref.py (seed 3, two public
inputs); the fixture records the circuit name (it carries the verifying key) and the input BITS so that the GPU test can feed
the evaluator side too.  Same pinning caveat as make_golden.py: these pin the HIP path to the oracle."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402


def case(spec, seed, extra=None):
    r = o.garble(spec, seed, capture_ct=False)
    d = {"circuit": spec, "program": spec, "replays": 1, "seed": seed,
         "gates": int(r.gate_counts.sum()), "n_ciphertexts": int(r.n_ciphertexts), "gate_counts": [int(x) for x in r.gate_counts],
         "delta": r.delta.tobytes().hex(), "ct_hash": r.ct_hash.tobytes().hex(),
         "output_label0_sha256": hashlib.sha256(r.output_label0.tobytes()).hexdigest(), "first_output_label0": r.output_label0[0].tobytes().hex()}
    d.update(extra or {})
    return d


def main():
    which = sys.argv[1]
    if which == "miller_loop":
        d = case("miller_loop", 9)
    elif which == "groth16_verify":
        import groth16_ref as G
        inst = G.make_instance(n_pub=2, seed=3)
        bits = G.input_bits(inst)
        d = case(G.circuit_name(inst), 11, {"instance": "groth16_ref.make_instance(n_pub=2, seed=3)", "input_bits_hex": bytes(__import__("numpy").packbits(bits, bitorder="little")).hex(),
                                           "n_inputs": int(bits.size), "expected_output": 1})
    elif which == "groth16_verify_compressed":
        import groth16_ref as G
        inst = G.make_instance(n_pub=2, seed=3)
        bits = G.compressed_input_bits(inst)
        d = case(G.compressed_circuit_name(inst), 12, {"instance": "groth16_ref.make_instance(n_pub=2, seed=3), compressed", "input_bits_hex": bytes(__import__("numpy").packbits(bits, bitorder="little")).hex(),
                                                      "n_inputs": int(bits.size), "expected_output": 1, "tamper_bit": 2 * 254 + 254})
    elif which == "groth16_verify_compressed_1pub":
        # the reference's own benchmark configuration: ONE public input (examples/groth16_garble.rs:107-110, groth16_cut_and_choose.rs:116-119)
        import groth16_ref as G
        inst = G.make_instance(n_pub=1, seed=6)
        bits = G.compressed_input_bits(inst)
        d = case(G.compressed_circuit_name(inst), 13, {"instance": "groth16_ref.make_instance(n_pub=1, seed=6), compressed", "input_bits_hex": bytes(__import__("numpy").packbits(bits, bitorder="little")).hex(),
                                                      "n_inputs": int(bits.size), "expected_output": 1, "tamper_bit": 254 + 254})
    else:
        raise SystemExit("unknown fixture: " + which)
    with open(os.path.join(HERE, which + "_golden.json"), "w") as f:
        json.dump(d, f)
    print({k: v for k, v in d.items() if k not in ("circuit", "program", "input_bits_hex")})


if __name__ == "__main__":
    main()
