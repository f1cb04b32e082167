"""Generates tests/golden/garble_golden.json from the CPU oracle (oracle/gsv_oracle.cpp).

The reference holds no golden ciphertext/label/hash literal and cannot be built here
(no Rust toolchain), so these fixtures pin the HIP path to the ORACLE, and the oracle itself is pinned
by FIPS-197 / SURVEY Appendix B vectors and the reference's property tests (tests/test_oracle_*.py).
First contact with a real `cargo` should confirm the fq12_mul case with AesNiHasher +
AESAccumulatingHash, seed 0 (tests/fq12_mul_e2e.rs shape).

Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as o  # noqa: E402

CASES = [
    # (oracle circuit spec, engine program spec, replays, seed)
    ("u254_add", "u254_add", 1, 0),
    ("driver_mix", "driver_mix", 1, 5),
    ("fq_mul", "fq_mul", 1, 0),
    ("fq_complex", "fq_complex", 1, 99),
    ("fq12_mul", "fq12_mul", 1, 0),
    ("fq12_mul_chain:3", "fq12_mul", 3, 1),
    ("fq12_square", "fq12_square", 1, 2),
    ("fq12_cyclotomic_square", "fq12_cyclotomic_square", 1, 3),
    ("fq12_sqmul_chain:2", "fq12_sqmul", 2, 4),
    ("fq_inverse", "fq_inverse", 1, 5),
    ("fq12_inverse", "fq12_inverse", 1, 6),
    ("fq12_frobenius:1", "fq12_frobenius:1", 1, 7),
    ("final_exp", "final_exp", 1, 8),  # 3,519,328,217 gates: the oracle needs ~2.5 minutes for this one
]


def main():
    out = {"generator": "tests/golden/make_golden.py (CPU oracle)", "cases": []}
    for spec, prog, replays, seed in CASES:
        r = o.garble(spec, seed, capture_ct=False)
        out["cases"].append({
            "circuit": spec, "program": prog, "replays": replays, "seed": seed,
            "gates": int(r.gate_counts.sum()), "n_ciphertexts": int(r.n_ciphertexts),
            "gate_counts": [int(x) for x in r.gate_counts],
            "delta": r.delta.tobytes().hex(),
            "ct_hash": r.ct_hash.tobytes().hex(),
            "output_label0_sha256": hashlib.sha256(r.output_label0.tobytes()).hexdigest(),
            "first_output_label0": r.output_label0[0].tobytes().hex(),
        })
        print(out["cases"][-1])
    with open(os.path.join(HERE, "garble_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
