#!/usr/bin/env python3
"""Per-step latency probe: times narrow programs (u254_add: 760 steps of 1-2 gates; fq_mul: 2812 steps, mean 147
gates) replayed many times on one instance, optionally under GSV_DIAG ablations (those need the diagnostic library:
`python garbled_snark_verifier_amd/build.py --diag`, then GSV_ENGINE_SO=.../libgsv_engine_diag.so).  Diagnostic tool, not a benchmark."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import garbled_snark_verifier_amd as gsv

eng = gsv.Engine(0)
SPECS = (("u254_add", 2000), ("fq_mul", 200), ("fq12_mul", 2))
if os.environ.get("GSV_ONLY"):
    SPECS = tuple(x for x in SPECS if x[0] == os.environ["GSV_ONLY"])
for spec, reps in SPECS:
    prog = gsv.Program.from_circuit(spec, chain_feedback=True)
    n_in = prog.info["n_inputs"]
    for B in ((1,) if os.environ.get("GSV_ONLY") else (1, 64)):
        d, f, t, inp = gsv.labels_from_seed(1, n_in)
        sess = gsv.Session(eng, prog, B, reps, 1)
        sess.set_garble_inputs(np.tile(d, (B, 1)), np.tile(np.stack([f, t]), (B, 1, 1)), np.tile(inp, (B, 1, 1)))
        for _ in range(2):
            sess.garble(0)
            sess.sync()
        ms = sess.last_kernel_ms()
        steps = prog.info["n_steps"]
        print("%-9s B=%-3d diag=%s: %.3f ms/replay, %.3f us/step (%d steps, %d and-steps, %d gates)" % (
            spec, B, os.environ.get("GSV_DIAG", "0"), ms / reps, ms / reps / steps * 1e3, steps, prog.info["n_and_steps"], prog.info["n_gates"]))
        sess.close()
