"""Generates cc16_verifier_golden.json: BASELINE config 5 at its REAL size — the GarbledInstanceCommit records of the 16 instances
of master seed 1234 on the FULL one-public-input `groth16_verify_compressed` circuit (11 456 865 898 gates each), every one garbled
by the CPU oracle from the flat gate stream (≈14 min of one core per instance; `-j` worker processes, each result cached under
/tmp so an interrupted run resumes).

  python tests/golden/make_cc16_verifier_golden.py [-j 8]

The GPU test (`tests/test_gpu_parity.py::test_cc16_verifier_full_size_on_one_gpu`) and `bench.py` (cc16_one_gpu, the e2e pass)
compare the records / ciphertext commitments the engine produces against this file, record by record."""
import hashlib
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402

MASTER, TOTAL = 1234, 16  # cut_and_choose/tests.rs:102: ChaCha20Rng::seed_from_u64(1234)
CACHE = "/tmp/gsv_cc16_verifier_cache"


def _circuit():
    return json.load(open(os.path.join(HERE, "groth16_verify_compressed_1pub_golden.json")))["circuit"]


def one(args):
    i, seed = args
    path = os.path.join(CACHE, "inst_%02d_%d.json" % (i, seed))
    if os.path.exists(path):
        return json.load(open(path))
    import oracle_lib as o
    from garbled_snark_verifier_amd import sharding
    g = o.garble(_circuit(), int(seed), capture_ct=False)
    rec = sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0)
    d = {"index": i, "seed": int(seed), "gates": int(g.gate_counts.sum()), "n_ciphertexts": int(g.n_ciphertexts), "delta": g.delta.tobytes().hex(),
         "ct_hash": g.ct_hash.tobytes().hex(), "first_output_label0": g.output_label0[0].tobytes().hex(), "record_len": int(rec.size),
         "record_sha256": hashlib.sha256(rec.tobytes()).hexdigest(), "record_hex": rec.tobytes().hex()}
    with open(path + ".tmp", "w") as f:
        json.dump(d, f)
    os.replace(path + ".tmp", path)
    return d


def main():
    from garbled_snark_verifier_amd import sharding
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 8
    os.makedirs(CACHE, exist_ok=True)
    # the seeds as the reference draws them (garbler.rs:201-203: rng.gen::<u64>() per instance), from the ORACLE's ChaCha stream — the
    # product's sharding.instance_seeds must reproduce them (tests/test_distributed_cpu.py)
    import oracle_lib as _o
    seeds = sharding.u64_stream_from_labels(_o.chacha_labels(MASTER, (TOTAL + 1) // 2), TOTAL)
    with mp.get_context("spawn").Pool(j) as pool:
        res = pool.map(one, [(i, int(seeds[i])) for i in range(TOTAL)], chunksize=1)
    res.sort(key=lambda d: d["index"])
    table = b"".join(bytes.fromhex(d["record_hex"]) for d in res)
    out = {"circuit_fixture": "groth16_verify_compressed_1pub_golden.json", "master_seed": MASTER, "total": TOTAL, "record_len": res[0]["record_len"],
           "gates": res[0]["gates"], "n_ciphertexts": res[0]["n_ciphertexts"], "seeds": [d["seed"] for d in res], "deltas": [d["delta"] for d in res],
           "ct_hashes": [d["ct_hash"] for d in res], "first_output_label0": [d["first_output_label0"] for d in res],
           "record_sha256": [d["record_sha256"] for d in res], "table_sha256": hashlib.sha256(table).hexdigest()}
    assert all(d["gates"] == out["gates"] and d["n_ciphertexts"] == out["n_ciphertexts"] for d in res)
    with open(os.path.join(HERE, "cc16_verifier_golden.json"), "w") as f:
        json.dump(out, f)
    print({k: v for k, v in out.items() if k not in ("seeds", "deltas", "ct_hashes", "record_sha256", "first_output_label0")})


if __name__ == "__main__":
    main()
