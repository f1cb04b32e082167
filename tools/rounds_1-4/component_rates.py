#!/usr/bin/env python3
"""Per-component device rates (one flat program each, 512 / 256 instances).  Diagnostic tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for spec in ("fq12_mul", "fq12_square", "fq12_cyclotomic_square", "fq12_inverse", "fq_inverse", "fq12_frobenius:1", "fq_mul"):
    prog = gsv.Program.from_circuit(spec)
    n_in = prog.info["n_inputs"]
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    sess = gsv.Session(eng, prog, B, 1, 1)
    sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
    for _ in range(3):
        sess.garble(0)
        sess.sync()
    ms = sess.last_kernel_ms()
    i = prog.info
    print("%-24s %11d gates  %6d steps (AND depth %6d)  %5.1f%% non-free  %8.2f ms  %.3e gates/s  %.2f us/step" % (
        spec, i["n_gates"], i["n_steps"], i["and_depth"], 100.0 * i["n_ciphertexts"] / i["n_gates"], ms, B * i["n_gates"] / ms * 1e3, ms * 1e3 / max(1, i["n_steps"])))
    sess.close()
