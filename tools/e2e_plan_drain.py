#!/usr/bin/env python3
"""The commitment-inclusive pass on a mid-sized plan (the Miller loop, 3.0e9 gates): garble B instances window by window while the host
drains every ciphertext over PCIe into the per-instance CBC-MACs.  Prints the engine's own account of where the pipeline waits
(GSV_DRAIN_STATS) for each setting of the drain knobs given as KEY=VALUE,... arguments.  Diagnostic tool.
usage: e2e_plan_drain.py instances [GSV_DRAIN_COPIES=8,GSV_DRAIN_CHUNK_MB=32 ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import garbled_snark_verifier_amd as gsv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
os.environ["GSV_DRAIN_STATS"] = "1"
eng = gsv.Engine(0)
case = json.load(open(os.path.join(ROOT, "tests", "golden", "miller_loop_golden.json")))
t0 = time.time()
plan = gsv.Plan.from_circuit(case["circuit"], bench.VERIFIER_UNITS, window_div=4)
print("plan: %d calls, %.3e gates, %.3e ciphertexts, built in %.1f s" % (plan.info["n_calls"], plan.info["n_gates"], plan.info["n_ciphertexts"], time.time() - t0), flush=True)
for cfg in sys.argv[2:] or [""]:
    kv = dict(x.split("=") for x in cfg.split(",") if x)
    for k in ("GSV_DRAIN_COPIES", "GSV_DRAIN_CHUNK_MB", "GSV_MAC_THREADS"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in kv.items() if k != "GSV_MAC_THREADS"})
    w = bench.VerifierWork(gsv, eng, plan, B, [case["seed"]] + list(range(900, 900 + B - 1)))
    t = w.run_pass(commit=True, threads=int(kv.get("GSV_MAC_THREADS", 0)))
    ok = w.ct_hashes[0].hex() == case["ct_hash"]
    print("%-50s B=%d: %.2f s -> %.3e gates/s, %.1f GB/s of ciphertexts, instance 0 MAC == fixture: %s" % (cfg or "(defaults)", B, t, B * plan.info["n_gates"] / t,
                                                                                                 B * plan.info["n_ciphertexts"] * 16e-9 / t, ok), flush=True)
    w.close()
