"""Generates cc16_golden.json: the GarbledInstanceCommit table of BASELINE config 5 on a SHORTENED circuit (Fq12 multiplication),
built from the CPU oracle's garblings — 16 seeds from master seed 1234 exactly as sharding.cut_and_choose_commit draws them.
bench.py --workload cc16 compares the table its ranks gather (GPU garbling, ciphertext commitments, one all-gather) with the
sha256 stored here; tests/test_gpu_parity.py compares record by record.

  python tests/golden/make_cc16_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import oracle_lib as o  # noqa: E402
from garbled_snark_verifier_amd import sharding  # noqa: E402

CIRCUIT, MASTER, TOTAL = "fq12_mul", 1234, 16  # cut_and_choose/tests.rs:102: ChaCha20Rng::seed_from_u64(1234)


def main():
    # the seeds as the reference draws them (garbler.rs:201-203: rng.gen::<u64>() per instance), from the ORACLE's ChaCha stream — the
    # product's sharding.instance_seeds must reproduce them (tests/test_distributed_cpu.py)
    import oracle_lib as _o
    seeds = sharding.u64_stream_from_labels(_o.chacha_labels(MASTER, (TOTAL + 1) // 2), TOTAL)
    recs = []
    for i in range(TOTAL):
        g = o.garble(CIRCUIT, int(seeds[i]), capture_ct=False)
        recs.append(sharding.commit_record(i, g.ct_hash.tobytes(), g.output_label0, g.delta, g.false_label0, g.true_label0, g.input_label0))
    table = np.stack(recs)
    d = {"circuit": CIRCUIT, "master_seed": MASTER, "total": TOTAL, "record_len": int(table.shape[1]), "seeds": [int(s) for s in seeds],
         "table_sha256": hashlib.sha256(table.tobytes()).hexdigest(), "ct_hashes": [bytes(r[8:24]).hex() for r in table],
         "record_sha256": [hashlib.sha256(r.tobytes()).hexdigest() for r in table]}
    with open(os.path.join(HERE, "cc16_golden.json"), "w") as f:
        json.dump(d, f)
    print({k: v for k, v in d.items() if k not in ("seeds", "ct_hashes", "record_sha256")})


if __name__ == "__main__":
    main()
