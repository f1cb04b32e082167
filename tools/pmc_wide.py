#!/usr/bin/env python3
"""The wide shape alone (fq12_mix as a plan of Fq12-level units, 1 024 instances, four per workgroup), three passes: the workload of
tools/pmc_wide.sh's counter passes (what does the vector-memory path do while the AES saturates LDS / VALU?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import garbled_snark_verifier_amd as gsv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = gsv.Engine(0)
# PMC_SHAPE: wide (default) | ladder (fq_sqrt as exp_chunk units) | inverse (fq_inverse as its three long calls) — tools/kernel_ab3.py's shapes
SHAPES = {"wide": ("fq12_mix", ["fq12::mul_montgomery", "fq12::square_montgomery"]), "ladder": ("fq_sqrt", ["fp254::exp_chunk"]),
          "inverse": ("fq_inverse", ["inverse::iteration_group", "inverse::divide_chains"])}
spec, units = SHAPES[os.environ.get("PMC_SHAPE", "wide")]
plan = gsv.Plan.from_circuit(spec, units, window_div=4)
d, f, t, inp = gsv.labels_from_seed(3, plan.info["n_inputs"])
sess = gsv.Session(eng, plan, B, retain_stream=False, concurrent_calls=1)
for _ in range(3):
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    t0 = time.perf_counter()
    sess.garble_streaming(discard=True)
    dt = time.perf_counter() - t0
    print("%.1f ms -> %.3e gates/s" % (dt * 1e3, B * plan.info["n_gates"] / dt), flush=True)
sess.close(); plan.close(); eng.close()
