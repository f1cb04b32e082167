#!/usr/bin/env python3
"""Device rates of the other kernel variants (evaluate, Blake3Hasher) on the bench program.  Diagnostic tool."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
prog = gsv.Program.from_circuit("fq12_mul", chain_feedback=True)
n_in, gates = prog.info["n_inputs"], prog.info["n_gates"]
R = 4
for B in (256, 512):
    d, f, t, inp = gsv.labels_from_seed(1, n_in)
    D, K, I = np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1))
    for hasher in ("aes", "blake3"):
        sess = gsv.Session(eng, prog, B, R, R)
        sess.set_hasher(hasher)
        sess.set_garble_inputs(D, K, I)
        for _ in range(2):
            sess.garble(0)
            sess.sync()
        g_ms = sess.last_kernel_ms()
        bits = np.zeros((B, n_in), np.uint8)
        ka = K.copy()
        ka[:, 1] ^= D  # evaluator holds true.label1
        sess.set_evaluate_inputs(ka, I, bits)
        for _ in range(2):
            sess.evaluate(0)
            sess.sync()
        e_ms = sess.last_kernel_ms()
        print("B=%d %-6s garble %.3e gates/s   evaluate %.3e gates/s" % (B, hasher, B * R * gates / g_ms * 1e3, B * R * gates / e_ms * 1e3))
        sess.close()
